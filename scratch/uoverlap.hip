// Can a wave's VALU work hide under the OTHER wave's MFMAs when both waves of a SIMD run one dependent MFMA chain each
// (the tile-major order of k_mfma_lp's gates) with K VALU instructions between consecutive MFMAs?
// hipcc --offload-arch=gfx950 -O2 uoverlap.hip -o uoverlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K, int NA>
__device__ __forceinline__ float chain(float r, int iters)
{
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = r, b = r + 1.f, x = r;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) x = __builtin_fmaf(x, 0.999f, 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int i = 0; i < NA; ++i) r += acc[i].x;
    return r + x;
}
__global__ __launch_bounds__(512) void k(int mode, int nact, long long* cyc, float* sink, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool act = wave == 0 || (nact == 2 && wave == 4);
    __syncthreads();
    long long c0 = __builtin_readcyclecounter();
    float r = lane * 0.001f;
    if (act) {
        switch (mode) {
        case 0: r = chain<0, 1>(r, iters); break;
        case 1: r = chain<2, 1>(r, iters); break;
        case 2: r = chain<4, 1>(r, iters); break;
        case 3: r = chain<8, 1>(r, iters); break;
        case 4: r = chain<12, 1>(r, iters); break;
        case 5: r = chain<16, 1>(r, iters); break;
        case 6: r = chain<0, 2>(r, iters); break;
        case 7: r = chain<8, 2>(r, iters); break;
        case 8: r = chain<16, 2>(r, iters); break;
        }
    }
    long long c1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = c1 - c0;
    sink[threadIdx.x] = r;
}
int main()
{
    long long* cyc; float* sink;
    hipMalloc(&cyc, 8 * 8); hipMalloc(&sink, 512 * 4);
    const int iters = 2000;
    const char* names[] = {"K=0 NA=1", "K=2 NA=1", "K=4 NA=1", "K=8 NA=1", "K=12 NA=1", "K=16 NA=1", "K=0 NA=2", "K=8 NA=2", "K=16 NA=2"};
    const int na[] = {1, 1, 1, 1, 1, 1, 2, 2, 2};
    for (int nact = 1; nact <= 2; ++nact)
        for (int mode = 0; mode < 9; ++mode) {
            long long h[8];
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, mode, nact, cyc, sink, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("%d wave(s) on the SIMD, %-10s: %.1f cycles per step of a wave (%d MFMA + K VALU); wave4 %.1f\n", nact, names[mode],
                   (double)h[0] / (iters * 8), na[mode], (double)h[4] / (iters * 8));
        }
    return 0;
}
