// What does the shader clock do under load? clock64() (s_memtime: shader cycles) against wall_clock64()
// (s_memrealtime: constant 100 MHz) around (a) back-to-back fp32 MFMAs on every SIMD, (b) dependent fp32 FMAs with one
// wave per SIMD (the cfg2 regime), (c) the same with 4 waves per SIMD. hipcc --offload-arch=gfx950 -O2 uclock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Rec { unsigned long long c0, c1, w0, w1; };
template <int MODE>
__global__ void k(Rec* out, float* sink, int iters)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, v = a;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v = __builtin_fmaf(v, b, a);
        } else {                               // four independent chains
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v = __builtin_fmaf(v, b, a); acc0.x = __builtin_fmaf(acc0.x, b, a);
                acc1.x = __builtin_fmaf(acc1.x, b, a); acc2.x = __builtin_fmaf(acc2.x, b, a);
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{c0, c1, w0, w1};
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc0.x + acc1.y + acc2.z + acc3.w + v;
}
template <int MODE>
void run(const char* name, int blocks, int threads, int iters, double flops_per_iter_per_wave)
{
    Rec* d; float* s;
    hipMalloc(&d, blocks * sizeof(Rec)); hipMalloc(&s, (size_t)blocks * threads * sizeof(float));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> h(blocks);
    hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    double ghz = 0, us = 0;
    for (auto& r : h) { ghz += (double)(r.c1 - r.c0) / ((double)(r.w1 - r.w0) * 10.0); us += (double)(r.w1 - r.w0) / 100.0; }
    ghz /= blocks; us /= blocks;
    const double waves = (double)blocks * threads / 64;
    printf("%-44s %5d wgs x %4d thr: kernel %8.1f us, per-wg %8.1f us, clock64/wall = %.3f GHz, %7.2f TFLOP/s\n", name, blocks, threads,
           ms * 1e3, us, ghz, waves * iters * flops_per_iter_per_wave / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(s);
}
int main()
{
    // 16x16x4 f32 MFMA = 2*16*16*4 = 2048 flop; 4 per iteration
    run<0>("mfma f32 16x16x4, 2 waves/SIMD, all CUs", 256, 512, 20000, 4 * 2048.0);
    run<0>("mfma f32 16x16x4, 1 wave/SIMD, all CUs", 256, 256, 20000, 4 * 2048.0);
    run<0>("mfma f32 16x16x4, 1 wave/SIMD, 32 CUs", 32, 256, 20000, 4 * 2048.0);
    run<1>("dependent fp32 FMA, 1 wave/SIMD, all CUs", 1024, 64, 20000, 16 * 2.0 * 64);
    run<1>("dependent fp32 FMA, 4 waves/SIMD, all CUs", 1024, 256, 20000, 16 * 2.0 * 64);
    run<1>("dependent fp32 FMA, 8 waves/SIMD, all CUs", 1024, 512, 20000, 16 * 2.0 * 64);
    run<2>("4 independent fp32 FMA chains, 1 wave/SIMD", 1024, 64, 20000, 16 * 2.0 * 64);
    run<2>("4 independent fp32 FMA chains, 2 waves/SIMD", 1024, 128, 20000, 16 * 2.0 * 64);
    run<2>("4 independent fp32 FMA chains, 4 waves/SIMD", 1024, 256, 20000, 16 * 2.0 * 64);
    run<2>("4 independent fp32 FMA chains, 8 waves/SIMD", 1024, 512, 20000, 16 * 2.0 * 64);
    return 0;
}
