// Do MFMA and VALU from two waves of one SIMD overlap? Which waves of a workgroup share a SIMD?
// How many independent accumulators saturate v_mfma_f32_16x16x4_f32?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NA> __device__ __forceinline__ float mfma_loop(float r, int iters)
{
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = r, b = r + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) r += acc[i].x + acc[i].y;
    return r;
}
// MFMA with one independent VALU op between consecutive MFMAs
template <int NA> __device__ __forceinline__ float mfma_valu_loop(float r, int iters, int nv)
{
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = r, b = r + 1.f, x = r, y = r * 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                x = __builtin_fmaf(x, 0.999f, 0.001f);
                if (nv >= 2) y = __builtin_fmaf(y, 0.998f, 0.002f);
                if (nv >= 4) { x = __builtin_fmaf(x, 0.997f, 0.003f); y = __builtin_fmaf(y, 0.996f, 0.004f); }
            }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) r += acc[i].x + acc[i].y;
    return r + x + y;
}
// role per wave: 0 idle, 1..8 MFMA loop with that many accumulators, 20 VALU loop, 31/32/34: 6-acc MFMA + 1/2/4 VALU each
__global__ __launch_bounds__(512) void k(const int* roles, long long* cyc, int* simd, float* sink, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = roles[wave];
    unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    __syncthreads();
    long long c0 = __builtin_readcyclecounter();
    float r = lane * 0.001f;
    switch (role) {
    case 1: r = mfma_loop<1>(r, iters); break;
    case 2: r = mfma_loop<2>(r, iters); break;
    case 3: r = mfma_loop<3>(r, iters); break;
    case 4: r = mfma_loop<4>(r, iters); break;
    case 5: r = mfma_loop<5>(r, iters); break;
    case 6: r = mfma_loop<6>(r, iters); break;
    case 8: r = mfma_loop<8>(r, iters); break;
    case 31: r = mfma_valu_loop<6>(r, iters, 1); break;
    case 32: r = mfma_valu_loop<6>(r, iters, 2); break;
    case 34: r = mfma_valu_loop<6>(r, iters, 4); break;
    case 20: {
        float x = r;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 24; ++j) x = __builtin_fmaf(x, 0.999f, 0.001f);
        }
        r = x;
    } break;
    default: break;
    }
    long long c1 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[wave] = c1 - c0; simd[wave] = (int)hwid; }
    sink[threadIdx.x] = r;
}
int main()
{
    int* roles; long long* cyc; int* simd; float* sink;
    hipMallocManaged(&roles, 8 * 4); hipMallocManaged(&cyc, 8 * 8); hipMallocManaged(&simd, 8 * 4); hipMalloc(&sink, 512 * 4);
    const int iters = 2000;
    const int cases[][8] = {
        {1,0,0,0,0,0,0,0}, {2,0,0,0,0,0,0,0}, {3,0,0,0,0,0,0,0}, {4,0,0,0,0,0,0,0}, {5,0,0,0,0,0,0,0}, {6,0,0,0,0,0,0,0}, {8,0,0,0,0,0,0,0},
        {20,0,0,0,0,0,0,0}, {6,0,0,0,20,0,0,0}, {6,20,0,0,0,0,0,0},
        {3,0,0,0,3,0,0,0}, {4,0,0,0,4,0,0,0}, {6,0,0,0,6,0,0,0},
        {31,0,0,0,0,0,0,0}, {32,0,0,0,0,0,0,0}, {34,0,0,0,0,0,0,0},
    };
    for (auto& c : cases) {
        for (int i = 0; i < 8; ++i) roles[i] = c[i];
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, roles, cyc, simd, sink, iters);
        hipDeviceSynchronize();
        printf("roles");
        for (int i = 0; i < 8; ++i) printf(" %2d", c[i]);
        printf("  cycles/iter:");
        for (int i = 0; i < 8; ++i) if (c[i]) printf(" w%d %7.1f", i, (double)cyc[i] / iters);
        printf("   simd:");
        for (int i = 0; i < 8; ++i) printf(" %d", (simd[i] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
