// lone-wave issue interval of v_mfma_f32_16x16x32_bf16: one dependent accumulation chain against two / four independent ones (one wave per SIMD; two waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int CHAINS>
__global__ void k(float* out, const float* in, int iters)
{
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{ in[threadIdx.x], in[threadIdx.x + 64], 0.f, 0.f };
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[128 + threadIdx.x + i]; b[i] = (__bf16)in[256 + threadIdx.x + i]; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[r % CHAINS], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[(blockIdx.x & 1023) * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + 64 * 1024)[0] = t1 - t0;
}
int main()
{
    float *d_in, *d_out;
    (void)hipMalloc(&d_in, 8192); (void)hipMalloc(&d_out, 64 * 1024 * 4 + 64);
    (void)hipMemset(d_in, 0, 8192);
    const int iters = 2000;
    for (int wps = 1; wps <= 2; ++wps)
        for (int chains = 1; chains <= 4; chains *= 2) {
            const int blocks = 256 * 4 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                if (chains == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters);
                else if (chains == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters);
                else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters);
                (void)hipDeviceSynchronize();
            }
            long long c; (void)hipMemcpy(&c, d_out + 64 * 1024, 8, hipMemcpyDeviceToHost);
            printf("v_mfma_f32_16x16x32_bf16, %d chain(s), %d wave(s) per SIMD: %.2f cycles per instruction\n", chains, wps, (double)c / (iters * 16.0));
        }
    return 0;
}
