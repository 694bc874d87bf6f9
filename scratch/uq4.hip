// Would FOUR streams per workgroup on v_mfma_f32_4x4x1_16b_f32 with the cell spread over the CU's four SIMDs beat the
// one-wave-per-stream LSTM-32 cell at cfg2's size (1024 streams = 4 per CU)? Models the per-frame critical loop only.
//   A   one wave per stream, 2 gate rows per lane, 66 FMAs per frame (today's recurrent wave in isolation)
//   Q4  4 waves x (8 units x 2 K-halves) : 17 MFMAs per frame and wave, halves added with permlane32 swaps, the
//       S=2 activation / exchange / cell update of LstmCell<32>, h through LDS, one s_barrier per frame
//   Q2  2 waves x 16 units (k_quad's arrangement): 35 MFMAs per frame and wave, lane-local cell update
//   S8  the two-wave split of the VALU cell with BOTH halves on one SIMD: 8 waves per workgroup, stream s on waves s and
//       s + 4 (16 units each, one gate row per lane: 33 FMAs), permlane32 + permlane16 exchange, one s_barrier per frame
// HELP > 0 adds helper waves that join every per-frame barrier with HELPV dependent VALU instructions of their own.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float sig(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.442695f)); }
__device__ __forceinline__ float tanh_r(float v)
{
    const float x = __builtin_fminf(__builtin_fmaxf(v, -7.9f), 7.9f), u = x * x;
    float p = -8.4887e-14f; p = __builtin_fmaf(p, u, 5.278e-11f); p = __builtin_fmaf(p, u, -2.0225e-08f); p = __builtin_fmaf(p, u, 1.11543e-05f);
    p = __builtin_fmaf(p, u, 0.0031039565f); p = __builtin_fmaf(p, u, 0.1308401f); p = __builtin_fmaf(p, u, 0.99999999f);
    float q = 0.00025461456f; q = __builtin_fmaf(q, u, 0.0244951795f); q = __builtin_fmaf(q, u, 0.4641733745f); q = __builtin_fmaf(q, u, 1.0f);
    return (p * x) * __builtin_amdgcn_rcpf(q);
}
struct Pair { float lo, hi; };
__device__ __forceinline__ Pair halves(float v)
{
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    unsigned a = r[0], b = r[1];
    return { __builtin_bit_cast(float, a), __builtin_bit_cast(float, b) };
}
// X = [xL | xH], Y = [yL | yH]  ->  low lanes xL + xH, high lanes yL + yH
__device__ __forceinline__ float fold(float x, float y)
{
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    unsigned a = r[0], b = r[1];
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

constexpr int HS = 36;          // floats per h row (stream)

// ---------------------------------------------------------------- A: one wave per stream
__global__ __launch_bounds__(64) void kA(const float* w, float* o, long long* t, int iters)
{
    __shared__ __attribute__((aligned(16))) float hb[2][HS];
    const int lane = threadIdx.x;
    float wr[2][36];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 36; ++i) wr[r][i] = w[(r * 36 + i) * 64 + lane];
    if (lane < HS) { hb[0][lane] = 0.01f * lane; hb[1][lane] = 0.f; }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    float c = 0.f, x = 0.1f;
    long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float4* hv = reinterpret_cast<const float4*>(hb[it & 1]);
        float acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[r] = wr[r][32] * x + wr[r][35];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float4 h = hv[k4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                acc[r] = __builtin_fmaf(wr[r][4 * k4], h.x, acc[r]); acc[r] = __builtin_fmaf(wr[r][4 * k4 + 1], h.y, acc[r]);
                acc[r] = __builtin_fmaf(wr[r][4 * k4 + 2], h.z, acc[r]); acc[r] = __builtin_fmaf(wr[r][4 * k4 + 3], h.w, acc[r]);
            }
        }
        const float a0 = sig(acc[0]);
        const float a1 = __builtin_fmaf(tanh_r(acc[1] * wr[0][33]), wr[0][34], wr[1][34]);
        const Pair p0 = halves(a0), p1 = halves(a1);
        c = __builtin_fmaf(p0.hi, c, p0.lo * p1.lo);
        const float hn = p1.hi * tanh_r(c);
        x = x * 0.999f + 0.001f;
        if (lane < 32) hb[(it + 1) & 1][lane] = hn;
        __builtin_amdgcn_wave_barrier();
    }
    long long c1 = __builtin_readcyclecounter();
    o[blockIdx.x * 64 + lane] = c + x;
    if (lane == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

// ---------------------------------------------------------------- Q4 / Q2
// NW recurrent waves; UPW units per wave; KH K-halves per wave (UPW * KH == 16 blocks)
template <int NW, int NACC, int HELP, int HELPV, int PRIO>
__global__ __launch_bounds__((NW + HELP) * 64) void kQ(const float* w, float* o, long long* t, int iters)
{
    constexpr int KH = NW == 4 ? 2 : 1;
    constexpr int UPW = 16 / KH;
    constexpr int KSTEPS = NW == 4 ? 17 : 33;
    __shared__ __attribute__((aligned(16))) float hb[2][4][HS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, j = lane & 3;
    const int q = KH == 2 ? (b >> 3) : 0, ul = KH == 2 ? (b & 7) : b;
    const int u = wave * UPW + ul;
    for (int i = tid; i < 2 * 4 * HS; i += (NW + HELP) * 64) (&hb[0][0][0])[i] = 0.001f * (i % 97);
    float wr[KSTEPS + 4];
    if (wave < NW) {
#pragma unroll
        for (int i = 0; i < KSTEPS + 4; ++i) wr[i] = w[(wave * 40 + i) * 64 + lane];
    }
    __syncthreads();
    float c = 0.f, x = 0.1f, hsum = 0.f;
    long long c0 = __builtin_readcyclecounter();
    if (wave < NW) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        const f32x4 bias = { wr[KSTEPS], wr[KSTEPS + 1], wr[KSTEPS + 2], wr[KSTEPS + 3] };
        for (int it = 0; it < iters; ++it) {
            const f32x4* hv = reinterpret_cast<const f32x4*>(&hb[it & 1][j][q * 16]);
            f32x4 acc[NACC];
            acc[0] = bias;
#pragma unroll
            for (int a = 1; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[KSTEPS - 1], x, acc[0], 0, 0, 0);
#pragma unroll
            for (int k4 = 0; k4 < (KSTEPS - 1) / 4; ++k4) {
                const f32x4 h4 = hv[k4];
                acc[(4 * k4 + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4], h4.x, acc[(4 * k4 + 1) % NACC], 0, 0, 0);
                acc[(4 * k4 + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4 + 1], h4.y, acc[(4 * k4 + 2) % NACC], 0, 0, 0);
                acc[(4 * k4 + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4 + 2], h4.z, acc[(4 * k4 + 3) % NACC], 0, 0, 0);
                acc[(4 * k4 + 4) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4 + 3], h4.w, acc[(4 * k4 + 4) % NACC], 0, 0, 0);
            }
            f32x4 g = acc[0];
#pragma unroll
            for (int a = 1; a < NACC; ++a) g += acc[a];
            float hn;
            if constexpr (NW == 4) {
                // low lanes: gates (i, g); high lanes: (f, o) — LstmCell<32>'s S = 2 arrangement
                const float v0 = fold(g.x, g.y);
                const float v1 = fold(g.z, g.w);
                const float a0 = sig(v0);
                const float a1 = __builtin_fmaf(tanh_r(v1 * wr[KSTEPS]), wr[KSTEPS + 1], wr[KSTEPS + 2]);
                const Pair p0 = halves(a0), p1 = halves(a1);
                c = __builtin_fmaf(p0.hi, c, p0.lo * p1.lo);
                hn = p1.hi * tanh_r(c);
                if (lane < 32) hb[(it + 1) & 1][j][u] = hn;
            } else {
                const float gi = sig(g.x), gf = sig(g.y), gg = tanh_r(g.z), go = sig(g.w);
                c = __builtin_fmaf(gf, c, gi * gg);
                hn = go * tanh_r(c);
                hb[(it + 1) & 1][j][u] = hn;
            }
            x = x * 0.999f + 0.001f;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        // helper waves: HELPV dependent VALU instructions per frame, one LDS read, the same barrier
        for (int it = 0; it < iters; ++it) {
            float a = hb[it & 1][j][lane >> 2];
#pragma unroll
            for (int v = 0; v < HELPV; ++v) a = __builtin_fmaf(a, 1.0001f, 1e-7f);
            hsum += a;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    long long c1 = __builtin_readcyclecounter();
    o[blockIdx.x * (NW + HELP) * 64 + tid] = c + x + hsum;
    if (tid == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

// ---------------------------------------------------------------- S8: VALU cell split over waves s, s + 4
__device__ __forceinline__ Pair rows16(float v)
{
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned a = r[0], b = r[1];
    return { __builtin_bit_cast(float, a), __builtin_bit_cast(float, b) };
}
template <int HELP, int HELPV, int PRIO>
__global__ __launch_bounds__((8 + HELP) * 64) void kS8(const float* w, float* o, long long* t, int iters)
{
    __shared__ __attribute__((aligned(16))) float hb[2][4][HS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = wave & 3, half = wave >> 2;              // stream, which 16 units
    for (int i = tid; i < 2 * 4 * HS; i += (8 + HELP) * 64) (&hb[0][0][0])[i] = 0.001f * (i % 97);
    float wr[37];
    if (wave < 8) {
#pragma unroll
        for (int i = 0; i < 37; ++i) wr[i] = w[((wave & 3) * 40 + i) * 64 + lane];
    }
    __syncthreads();
    float c = 0.f, x = 0.1f, hsum = 0.f;
    long long c0 = __builtin_readcyclecounter();
    if (wave < 8) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it < iters; ++it) {
            const float4* hv = reinterpret_cast<const float4*>(&hb[it & 1][j][0]);
            float acc = __builtin_fmaf(wr[32], x, wr[33]);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const float4 h = hv[k4];
                acc = __builtin_fmaf(wr[4 * k4], h.x, acc); acc = __builtin_fmaf(wr[4 * k4 + 1], h.y, acc);
                acc = __builtin_fmaf(wr[4 * k4 + 2], h.z, acc); acc = __builtin_fmaf(wr[4 * k4 + 3], h.w, acc);
            }
            // lane = gate * 16 + unit: every lane one activation in the common form, then both exchanges
            const float a = __builtin_fmaf(tanh_r(acc * wr[34]), wr[35], wr[36]);
            const Pair p = halves(a);
            const Pair lo = rows16(p.lo), hi = rows16(p.hi);
            c = __builtin_fmaf(lo.hi, c, lo.lo * hi.lo);
            const float hn = hi.hi * tanh_r(c);
            x = x * 0.999f + 0.001f;
            if (lane < 16) hb[(it + 1) & 1][j][half * 16 + lane] = hn;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            float a = hb[it & 1][lane & 3][lane >> 2];
#pragma unroll
            for (int v = 0; v < HELPV; ++v) a = __builtin_fmaf(a, 1.0001f, 1e-7f);
            hsum += a;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    long long c1 = __builtin_readcyclecounter();
    o[blockIdx.x * (8 + HELP) * 64 + tid] = c + x + hsum;
    if (tid == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

template <class F> void timeit(const char* name, F launch, long long* t, int iters)
{
    launch(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s %7.1f cycles/frame (wave 0 of block 0)   %8.1f us per 256 frames (whole launch)\n", name, (double)h / iters, ms * 1e3 * 256 / iters);
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    float* w; float* o; long long* t;
    const int NWT = 4 * 40 * 64;
    hipMalloc(&w, NWT * 4); hipMalloc(&o, 1024 * 640 * 4); hipMalloc(&t, 16);
    float* hw = new float[NWT];
    for (int i = 0; i < NWT; ++i) hw[i] = 0.01f * ((i * 7919) % 101 - 50) / 50.f;
    hipMemcpy(w, hw, NWT * 4, hipMemcpyHostToDevice);
    const int iters = 8000;
    for (int rep = 0; rep < reps; ++rep) {
        timeit("A  one wave per stream, 1024 blocks", [&] { hipLaunchKernelGGL(kA, dim3(1024), dim3(64), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q4 1 acc, 256 blocks", [&] { hipLaunchKernelGGL((kQ<4, 1, 0, 0, 0>), dim3(256), dim3(256), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q4 2 acc", [&] { hipLaunchKernelGGL((kQ<4, 2, 0, 0, 0>), dim3(256), dim3(256), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q4 2 acc + 2 helpers x 30 VALU", [&] { hipLaunchKernelGGL((kQ<4, 2, 2, 30, 0>), dim3(256), dim3(384), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q4 2 acc + 2 helpers x 30 VALU, prio", [&] { hipLaunchKernelGGL((kQ<4, 2, 2, 30, 1>), dim3(256), dim3(384), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q4 2 acc + 2 helpers x 60 VALU, prio", [&] { hipLaunchKernelGGL((kQ<4, 2, 2, 60, 1>), dim3(256), dim3(384), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q4 1 acc + 2 helpers x 30 VALU, prio", [&] { hipLaunchKernelGGL((kQ<4, 1, 2, 30, 1>), dim3(256), dim3(384), 0, 0, w, o, t, iters); }, t, iters);
        timeit("S8 VALU split, halves on one SIMD, 256 blocks", [&] { hipLaunchKernelGGL((kS8<0, 0, 0>), dim3(256), dim3(512), 0, 0, w, o, t, iters); }, t, iters);
        timeit("S8 + 2 helpers x 30 VALU, prio", [&] { hipLaunchKernelGGL((kS8<2, 30, 1>), dim3(256), dim3(640), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q2 2 acc (k_quad arrangement), 256 blocks", [&] { hipLaunchKernelGGL((kQ<2, 2, 0, 0, 0>), dim3(256), dim3(128), 0, 0, w, o, t, iters); }, t, iters);
        timeit("Q2 2 acc + 2 helpers x 30, prio", [&] { hipLaunchKernelGGL((kQ<2, 2, 2, 30, 1>), dim3(256), dim3(256), 0, 0, w, o, t, iters); }, t, iters);
    }
    return 0;
}
