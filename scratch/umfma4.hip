// v_mfma_f32_4x4x1_16b_f32: issue rate and dependent-accumulator latency (one wave, and two waves on one SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NA> __device__ __forceinline__ float loop(float r, int iters)
{
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = r, b = r + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 24 / NA; ++j)
#pragma unroll
            for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) r += acc[i].x + acc[i].y;
    return r;
}
__global__ __launch_bounds__(512) void k(const int* roles, long long* cyc, float* sink, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = roles[wave];
    __syncthreads();
    long long c0 = __builtin_readcyclecounter();
    float r = lane * 0.001f;
    switch (role) {
    case 1: r = loop<1>(r, iters); break;
    case 2: r = loop<2>(r, iters); break;
    case 3: r = loop<3>(r, iters); break;
    case 4: r = loop<4>(r, iters); break;
    case 6: r = loop<6>(r, iters); break;
    default: break;
    }
    long long c1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = c1 - c0;
    sink[threadIdx.x] = r;
}
int main()
{
    int* roles; long long* cyc; float* sink;
    hipMallocManaged(&roles, 32); hipMallocManaged(&cyc, 64); hipMalloc(&sink, 2048);
    const int iters = 2000;
    const int cases[][8] = { {1,0,0,0,0,0,0,0}, {2,0,0,0,0,0,0,0}, {3,0,0,0,0,0,0,0}, {4,0,0,0,0,0,0,0}, {6,0,0,0,0,0,0,0},
                             {1,0,0,0,1,0,0,0}, {2,0,0,0,2,0,0,0}, {4,0,0,0,4,0,0,0} };
    for (auto& c : cases) {
        for (int i = 0; i < 8; ++i) roles[i] = c[i];
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, roles, cyc, sink, iters);
        hipDeviceSynchronize();
        printf("accumulators per wave:");
        for (int i = 0; i < 8; ++i) if (c[i]) printf(" w%d=%d -> %6.2f cycles/MFMA", i, c[i], (double)cyc[i] / iters / 24);
        printf("\n");
    }
    return 0;
}
