// Round 4, follow-up to uoverlap2.hip: is the missing MFMA / VALU overlap a property of v_mfma_f32_16x16x4_f32 with its
// accumulator in VGPRs, or of the machine? Same "fill" experiment (one wave per SIMD, F independent FMAs after every MFMA,
// three accumulators round-robin), varying the matrix instruction and where its accumulator lives:
//   f32 16x16x4 acc in VGPRs (builtin)      | f32 16x16x4 acc in AGPRs (inline asm, "a" constraint)
//   f32 32x32x2 (16 passes, 64 cycles)      | bf16 32x32x16 (the instruction MI355X_MICROARCH.md measured fillers under)
//   bf16 16x16x32
// Two-wave numbers pick the partner wave by HW_ID (the wave that really shares wave 0's SIMD).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize uoverlap3.hip -o uoverlap3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int KIND> struct Acc { typedef f32x4 T; };
template <> struct Acc<2> { typedef f32x16 T; };
template <> struct Acc<3> { typedef f32x16 T; };

template <int KIND, int F>
__device__ __forceinline__ float fill(float r, int iters)
{
    typename Acc<KIND>::T acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = r + i;
    float a = r, b = r + 1.f;
    bf16x8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(r + i); hb[i] = (__bf16)(r - i); }
    float x[4] = {r, r + 1.f, r + 2.f, r + 3.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            auto& c = acc[j % 3];
            if constexpr (KIND == 0) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
            if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
            if constexpr (KIND == 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
            if constexpr (KIND == 3) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c, 0, 0, 0);
            if constexpr (KIND == 4) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c, 0, 0, 0);
            SB();
#pragma unroll
            for (int k = 0; k < F; ++k) { float& v = x[(j * F + k) & 3]; v = __builtin_fmaf(v, 0.999f, 0.001f); }
            SB();
        }
    }
    float s = x[0] + x[1] + x[2] + x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) s += acc[i][0];
    return s;
}

struct Args { int kind, f, nact, iters; };
__global__ __launch_bounds__(512) void k(Args A, long long* cyc, int* simd_out, float* sink)
{
    __shared__ int simd[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    if (lane == 0) simd[wave] = (hwid >> 4) & 3;
    __syncthreads();
    int partner = -1;
    for (int w = 7; w >= 1; --w) if (simd[w] == simd[0]) partner = w;
    const bool act = wave == 0 || (A.nact == 2 && wave == partner);
    const long long c0 = __builtin_readcyclecounter();
    float r = lane * 0.001f;
    if (act) {
#define CASE(KIND, F) if (A.kind == KIND && A.f == F) r = fill<KIND, F>(r, A.iters);
#define KINDS(F) CASE(0, F) CASE(1, F) CASE(2, F) CASE(3, F) CASE(4, F)
        KINDS(0) KINDS(1) KINDS(2) KINDS(4) KINDS(6) KINDS(8) KINDS(12)
    }
    const long long c1 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[partner > 0 && wave == partner ? 4 : partner > 0 && wave == 4 ? partner : wave] = c1 - c0; simd_out[wave] = simd[wave]; }      // the partner's time in column 4
    sink[threadIdx.x] = r;
}

int main()
{
    long long* cyc; float* sink; int* simd;
    (void)hipMalloc(&cyc, 64); (void)hipMalloc(&sink, 2048); (void)hipMalloc(&simd, 32);
    const char* names[] = {"f32 16x16x4, acc in VGPRs", "f32 16x16x4, acc in AGPRs", "f32 32x32x2", "bf16 32x32x16", "bf16 16x16x32"};
    const int fs[] = {0, 1, 2, 4, 6, 8, 12};
    const int iters = 1000;
    int hs[8];
    printf("# cycles per MFMA of the SIMD; F independent FMAs after every MFMA, three accumulators round-robin\n");
    for (int kind = 0; kind < 5; ++kind)
        for (int nact = 1; nact <= 2; ++nact) {
            printf("%-28s %d wave%s/SIMD:", names[kind], nact, nact == 1 ? " " : "s");
            for (int f : fs) {
                long long h[8];
                for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, Args{kind, f, nact, iters}, cyc, simd, sink); (void)hipDeviceSynchronize(); }
                (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
                (void)hipMemcpy(hs, simd, 32, hipMemcpyDeviceToHost);
                // both waves' MFMAs over the time of the LATER one (the SIMD favours its older wave: wave 0 alone would flatter it)
                const long long t = nact == 2 && h[4] > h[0] ? h[4] : h[0];
                printf("  F=%-2d %6.1f", f, (double)t / (iters * 12.0 * nact));
            }
            printf("\n");
        }
    printf("# SIMD of waves 0..7 in the last launch:");
    for (int w = 0; w < 8; ++w) printf(" %d", hs[w]);
    printf("\n");
    return 0;
}
