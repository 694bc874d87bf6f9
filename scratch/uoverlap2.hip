// Round 4: does VALU work hide beside MFMAs on one SIMD of gfx950 — measured WITHOUT the same-accumulator cliff that
// scratch/uoverlap.hip (round 3) ran into (one accumulator, K fillers between two DEPENDENT MFMAs: MI355X_MICROARCH.md
// documents that as +43 cycles for the first extra issue slot). Three experiments, v_mfma_f32_16x16x4_f32 throughout:
//   fill   : NA independent accumulators round-robin (as lp_gates issues them), F fillers between consecutive MFMAs on
//            DIFFERENT accumulators; filler kinds: independent FMAs, transcendentals (v_exp_f32), LDS reads;
//            one wave per SIMD and two. Reported: cycles per MFMA of the SIMD (floor: 32).
//   segment: two waves of a SIMD, each iteration = S MFMAs (3 accumulators) and V VALU (the LSTM cell update's mix);
//            in phase (both MFMA, then both VALU — what k_mfma_lp / k_gru_gm do today) against
//            anti-phase (A: MFMA while B: VALU, then swapped), each with a workgroup barrier per iteration.
//   tick   : the proposed tick of k_mfma_lp: N1 bare MFMAs (the recurrent half) then N2 MFMAs (the next frame's
//            below-half) with the cell update of three tiles interleaved by sched_group_barrier, against
//            N1 + N2 bare MFMAs followed by the cell update; eight waves, one barrier per tick.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize uoverlap2.hip -o uoverlap2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

enum { K_FMA = 0, K_EXP = 1, K_LDS = 2 };

template <int NA, int F, int KIND>
__device__ __forceinline__ float fill(float r, int iters, const float* lds, int lane)
{
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{r + i, 0, 0, 0};
    float a = r, b = r + 1.f;
    float x[4] = {r, r + 1.f, r + 2.f, r + 3.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            acc[j % NA] = MFMA(a, b, acc[j % NA]);
            SB();
#pragma unroll
            for (int k = 0; k < F; ++k) {
                float& v = x[(j * F + k) & 3];
                if (KIND == K_FMA) v = __builtin_fmaf(v, 0.999f, 0.001f);
                else if (KIND == K_EXP) v = __builtin_amdgcn_exp2f(v);
                else v += lds[64 * ((j * F + k) & 15) + lane];      // ds_read + dependent add (the wait lands before the add)
            }
            SB();
        }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) r += acc[i].x;
    return r + x[0] + x[1] + x[2] + x[3];
}

// the LSTM cell update of one tile as k_mfma_lp issues it: three sigmoids (exp2, add, rcp), two rational tanh
// (degree 3 over 3 in x^2: 13 instructions + rcp), the c / h arithmetic and one LDS write: ~43 instructions, 6 transcendental
__device__ __forceinline__ float sig(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v)); }
__device__ __forceinline__ float tanhr(float x)
{
    const float x2 = x * x;
    float p = __builtin_fmaf(x2, 2.1e-5f, 1.2e-3f);
    p = __builtin_fmaf(p, x2, 5.1e-2f);
    p = __builtin_fmaf(p, x2, 1.f);
    float q = __builtin_fmaf(x2, 1.1e-6f, 3.4e-4f);
    q = __builtin_fmaf(q, x2, 2.2e-2f);
    q = __builtin_fmaf(q, x2, 3.8e-1f);
    q = __builtin_fmaf(q, x2, 1.f);
    const float cl = __builtin_fminf(__builtin_fmaxf(x, -9.f), 9.f);
    return cl * p * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float cell(const f32x4& g, float& c)
{
    const float gi = sig(g.x), gf = sig(g.y), gg = tanhr(g.z), go = sig(g.w);
    c = __builtin_fmaf(gf, c, gi * gg);
    return go * tanhr(c);
}

// ---- segment: S MFMAs and the cell update of NT tiles per iteration; phase = 0: MFMA first, 1: VALU first
template <int S, int NTILE>
__device__ __forceinline__ float segment(float r, int iters, int phase, float* lds, int lane, int wave, bool barrier)
{
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{r + i, r, r, r};
    float a = r * 1e-3f, b = 1e-3f;
    float c[NTILE];
    f32x4 g[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) { c[t] = r; g[t] = f32x4{r, -r, r * 0.5f, r}; }
    auto mfmas = [&]() {
#pragma unroll
        for (int j = 0; j < S; ++j) acc[j % 3] = MFMA(a, b, acc[j % 3]);
    };
    auto valus = [&]() {
#pragma unroll
        for (int t = 0; t < NTILE; ++t) lds[(wave * NTILE + t) * 64 + lane] = cell(g[t], c[t]);
    };
    for (int it = 0; it < iters; ++it) {
        SB();
        if (phase == 0) { mfmas(); SB(); if (barrier) __builtin_amdgcn_s_barrier(); SB(); valus(); }
        else            { valus(); SB(); if (barrier) __builtin_amdgcn_s_barrier(); SB(); mfmas(); }
        SB();
        if (barrier) __syncthreads();
#pragma unroll
        for (int t = 0; t < NTILE; ++t) g[t] = acc[t % 3] * 1e-3f;      // the next update works on this iteration's results
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) s += c[t];
    return s + acc[0].x + acc[1].x + acc[2].x;
}

// ---- tick: N1 recurrent MFMAs (bare), then N2 early MFMAs of the next frame interleaved with the cell update of three tiles
template <int N1, int N2, bool INTERLEAVE, int VPER>
__device__ __forceinline__ float tick(float r, int iters, float* lds, int lane, int wave)
{
    f32x4 acc[3], nxt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i] = f32x4{r, r, r, r}; nxt[i] = f32x4{r + i, 0, 0, 0}; }
    float a = r * 1e-3f, b = 1e-3f, c[3] = {r, r, r};
    for (int it = 0; it < iters; ++it) {
        SB();
#pragma unroll
        for (int j = 0; j < N1; ++j) acc[j % 3] = MFMA(a, lds[64 * (j & 15) + lane], acc[j % 3]);
        SB();
        if (!INTERLEAVE) {
#pragma unroll
            for (int j = 0; j < N2; ++j) nxt[j % 3] = MFMA(b, a, nxt[j % 3]);
            SB();
        }
        float h[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) h[t] = cell(acc[t], c[t]);
        if (INTERLEAVE) {
#pragma unroll
            for (int j = 0; j < N2; ++j) nxt[j % 3] = MFMA(b, a, nxt[j % 3]);
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);   // VPER VALU (transcendentals count as VALU here)
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) lds[1024 + (wave * 3 + t) * 64 + lane] = h[t];
        SB();
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t) { acc[t] = nxt[t] * 1e-3f + h[t]; nxt[t] = f32x4{h[t], 0, 0, 0}; }
    }
    return c[0] + c[1] + c[2] + acc[0].x + acc[1].x + acc[2].x;
}


// the same cell update cut into eight slices of 4-6 instructions, so that the interleave is written out by hand
// (one slice behind each early MFMA, fenced by sched_barrier) instead of left to sched_group_barrier
struct CellSt { float ei, ef, eo, gi, gf, go, x2, p, q, gg, c, x; };
template <int S> __device__ __forceinline__ void cell_slice(CellSt& s, const f32x4& g)
{
    if constexpr (S == 0) { s.ei = __builtin_amdgcn_exp2f(g.x); s.ef = __builtin_amdgcn_exp2f(g.y); s.eo = __builtin_amdgcn_exp2f(g.w); s.x = g.z; s.x2 = g.z * g.z; }
    if constexpr (S == 1) { s.gi = __builtin_amdgcn_rcpf(1.f + s.ei); s.gf = __builtin_amdgcn_rcpf(1.f + s.ef); s.go = __builtin_amdgcn_rcpf(1.f + s.eo); }
    if constexpr (S == 2 || S == 5) { s.p = __builtin_fmaf(s.x2, 2.1e-5f, 1.2e-3f); s.p = __builtin_fmaf(s.p, s.x2, 5.1e-2f); s.p = __builtin_fmaf(s.p, s.x2, 1.f);
                                      s.q = __builtin_fmaf(s.x2, 1.1e-6f, 3.4e-4f); s.q = __builtin_fmaf(s.q, s.x2, 2.2e-2f); }
    if constexpr (S == 3 || S == 6) { s.q = __builtin_fmaf(s.q, s.x2, 3.8e-1f); s.q = __builtin_fmaf(s.q, s.x2, 1.f);
                                      const float cl = __builtin_fminf(__builtin_fmaxf(s.x, -9.f), 9.f); s.gg = cl * s.p * __builtin_amdgcn_rcpf(s.q); }
    if constexpr (S == 4) { s.c = __builtin_fmaf(s.gf, s.c, s.gi * s.gg); s.x = s.c; s.x2 = s.c * s.c; }
    if constexpr (S == 7) { s.gg = s.go * s.gg; }
}

template <int J, int N2, int STRIDE>
__device__ __forceinline__ void early(f32x4 (&nxt)[3], CellSt (&st)[3], const f32x4 (&acc)[3], float a, float b)
{
    if constexpr (J < N2) {
        nxt[J % 3] = MFMA(b, a, nxt[J % 3]);
        SB();
        if constexpr (J % STRIDE == 0 && J / STRIDE < 24) cell_slice<(J / STRIDE) % 8>(st[(J / STRIDE) / 8], acc[(J / STRIDE) / 8]);
        SB();
        early<J + 1, N2, STRIDE>(nxt, st, acc, a, b);
    }
}
template <int K0>
__device__ __forceinline__ void rest(CellSt (&st)[3], const f32x4 (&acc)[3])
{
    if constexpr (K0 < 24) { cell_slice<K0 % 8>(st[K0 / 8], acc[K0 / 8]); rest<K0 + 1>(st, acc); }
}
template <int N1, int N2, int STRIDE>
__device__ __forceinline__ float tick_manual(float r, int iters, float* lds, int lane, int wave)
{
    f32x4 acc[3], nxt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i] = f32x4{r, r, r, r}; nxt[i] = f32x4{r + i, 0, 0, 0}; }
    float a = r * 1e-3f, b = 1e-3f;
    CellSt st[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) st[t].c = r;
    for (int it = 0; it < iters; ++it) {
        SB();
#pragma unroll
        for (int j = 0; j < N1; ++j) acc[j % 3] = MFMA(a, lds[64 * (j & 15) + lane], acc[j % 3]);
        SB();
        // 24 slices (tile-major: tile 0's eight, then tile 1's ...), one behind every STRIDE-th early MFMA
        early<0, N2, STRIDE>(nxt, st, acc, a, b);
        // slices that did not fit behind an MFMA
        rest<(N2 + STRIDE - 1) / STRIDE>(st, acc);
#pragma unroll
        for (int t = 0; t < 3; ++t) lds[1024 + (wave * 3 + t) * 64 + lane] = st[t].gg;
        SB();
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t) { acc[t] = nxt[t] * 1e-3f + st[t].gg; nxt[t] = f32x4{st[t].gg, 0, 0, 0}; }
    }
    return st[0].c + st[1].c + st[2].c + acc[0].x + acc[1].x + acc[2].x;
}

struct Args { int exp, mode, sub, nact, iters; };
__global__ __launch_bounds__(512) void k(Args A, long long* cyc, float* sink)
{
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ int simd[8];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = 1e-3f * i;
    if (lane == 0) simd[wave] = (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) >> 4) & 3;      // HW_ID.simd_id
    __syncthreads();
    // which wave really shares wave 0's SIMD (the first run of this file assumed wave 4: not true on every launch), and
    // whether this wave is the second one on its SIMD
    int partner = -1, slot = 0;
    for (int w = 7; w >= 1; --w) if (simd[w] == simd[0]) partner = w;
    for (int w = 0; w < wave; ++w) if (simd[w] == simd[wave]) slot = 1;
    // nact = 1: wave 0 only; 2: wave 0 and its SIMD partner; 8: all
    const bool act = A.nact == 8 || wave == 0 || (A.nact == 2 && wave == partner);
    const long long c0 = __builtin_readcyclecounter();
    float r = lane * 0.001f;
    const int it = A.iters;
    if (A.exp == 0 && act) {
#define FILLCASE(ID, NA, F, KIND) case ID: r = fill<NA, F, KIND>(r, it, lds, lane); break;
        switch (A.mode) {
            FILLCASE(0, 3, 0, K_FMA) FILLCASE(1, 3, 1, K_FMA) FILLCASE(2, 3, 2, K_FMA) FILLCASE(3, 3, 3, K_FMA) FILLCASE(4, 3, 4, K_FMA)
            FILLCASE(5, 3, 5, K_FMA) FILLCASE(6, 3, 6, K_FMA) FILLCASE(7, 3, 8, K_FMA) FILLCASE(8, 3, 12, K_FMA)
            FILLCASE(10, 3, 1, K_EXP) FILLCASE(11, 3, 2, K_EXP) FILLCASE(12, 3, 3, K_EXP) FILLCASE(13, 3, 4, K_EXP)
            FILLCASE(20, 3, 1, K_LDS) FILLCASE(21, 3, 2, K_LDS)
            FILLCASE(30, 1, 0, K_FMA) FILLCASE(31, 1, 1, K_FMA) FILLCASE(32, 1, 2, K_FMA) FILLCASE(33, 1, 4, K_FMA)
            FILLCASE(40, 4, 0, K_FMA) FILLCASE(41, 4, 2, K_FMA) FILLCASE(42, 4, 4, K_FMA) FILLCASE(43, 4, 6, K_FMA)
            FILLCASE(50, 2, 0, K_FMA) FILLCASE(51, 2, 2, K_FMA) FILLCASE(52, 2, 4, K_FMA)
        }
    } else if (A.exp == 1) {
        // sub: 0 = in phase, 1 = anti-phase (waves >= 4 start with the other segment); mode: 0 = with barriers, 1 = free-running
        const int phase = A.sub == 1 && slot == 1 ? 1 : 0;
        if (act) r = segment<36, 3>(r, it, phase, lds + 2048, lane, wave, A.mode == 0);
        else if (A.mode == 0) for (int i = 0; i < 2 * it; ++i) __syncthreads();
    } else if (A.exp == 2) {
        switch (A.mode) {
        case 0: r = tick<72, 36, false, 0>(r, it, lds, lane, wave); break;
        case 1: r = tick<72, 36, true, 3>(r, it, lds, lane, wave); break;
        case 2: r = tick<72, 36, true, 4>(r, it, lds, lane, wave); break;
        case 3: r = tick<72, 36, true, 5>(r, it, lds, lane, wave); break;
        case 4: r = tick<72, 36, true, 2>(r, it, lds, lane, wave); break;
        case 5: r = tick<108, 0, false, 0>(r, it, lds, lane, wave); break;      // no early half at all: today's critical path
        case 6: r = tick_manual<72, 36, 1>(r, it, lds, lane, wave); break;      // a slice behind each of the first 24 early MFMAs
        case 7: r = tick_manual<72, 36, 2>(r, it, lds, lane, wave); break;      // ... behind every second one (18 fit, 6 behind the last MFMA)
        case 8: r = tick_manual<72, 48, 2>(r, it, lds, lane, wave); break;      // 48 early MFMAs, a slice behind every second
        case 9: r = tick_manual<72, 24, 1>(r, it, lds, lane, wave); break;      // 24 early MFMAs, a slice behind each
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + (partner > 0 && wave == partner ? 4 : partner > 0 && wave == 4 ? partner : wave)] = c1 - c0;      // (the partner's time is reported in column 4)
    sink[blockIdx.x * 512 + threadIdx.x] = r;
}

int main(int argc, char** argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 1;
    long long* cyc; float* sink;
    hipMalloc(&cyc, grid * 8 * 8); hipMalloc(&sink, grid * 512 * 4);
    auto run = [&](Args A, long long* h) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, A, cyc, sink); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    };
    long long h[8];
    printf("# fill: cycles per MFMA on one SIMD (floor 32); NA accumulators round-robin, F fillers after every MFMA\n");
    struct { int id, na, f; const char* kind; } fills[] = {
        {0, 3, 0, "fma"}, {1, 3, 1, "fma"}, {2, 3, 2, "fma"}, {3, 3, 3, "fma"}, {4, 3, 4, "fma"}, {5, 3, 5, "fma"}, {6, 3, 6, "fma"}, {7, 3, 8, "fma"}, {8, 3, 12, "fma"},
        {10, 3, 1, "exp"}, {11, 3, 2, "exp"}, {12, 3, 3, "exp"}, {13, 3, 4, "exp"}, {20, 3, 1, "lds"}, {21, 3, 2, "lds"},
        {30, 1, 0, "fma"}, {31, 1, 1, "fma"}, {32, 1, 2, "fma"}, {33, 1, 4, "fma"},
        {50, 2, 0, "fma"}, {51, 2, 2, "fma"}, {52, 2, 4, "fma"},
        {40, 4, 0, "fma"}, {41, 4, 2, "fma"}, {42, 4, 4, "fma"}, {43, 4, 6, "fma"}};
    const int iters = 1000;
    for (auto& f : fills) {
        double v[2];
        for (int nact = 1; nact <= 2; ++nact) {
            run(Args{0, f.id, 0, nact, iters}, h);
            // per SIMD: with two waves the SIMD issues 2 x 12 MFMAs per iteration of a wave
            // (both waves' MFMAs over the time of the LATER wave: the SIMD favours its older wave)
            v[nact - 1] = (double)(nact == 2 && h[4] > h[0] ? h[4] : h[0]) / (iters * 12.0 * nact);
        }
        printf("fill NA=%d F=%-2d %s : 1 wave/SIMD %6.1f   2 waves/SIMD %6.1f  cycles per MFMA\n", f.na, f.f, f.kind, v[0], v[1]);
    }
    printf("# segment: 36 MFMAs (3 accumulators) + cell update of 3 tiles (~130 VALU, 18 transcendental) per iteration and wave\n");
    for (int mode = 0; mode < 2; ++mode) {
        run(Args{1, mode, 0, 1, iters}, h);
        const double one = (double)h[0] / iters;
        run(Args{1, mode, 0, 2, iters}, h);
        const double inph = (double)h[0] / iters;
        run(Args{1, mode, 1, 2, iters}, h);
        const double anti = (double)h[0] / iters, anti4 = (double)h[4] / iters;
        printf("segment %s: one wave alone %7.0f | two waves in phase %7.0f | anti-phase %7.0f (wave 4: %7.0f) cycles per iteration (36 MFMAs = 1152 of pipe time per wave)\n",
               mode == 0 ? "barrier per segment" : "free-running      ", one, inph, anti, anti4);
    }
    printf("# tick: eight waves (two per SIMD), one barrier per tick; per wave 72 recurrent + 36 early MFMAs + cell update of 3 tiles\n");
    const char* tn[] = {"108 bare MFMAs, then the cell update (today)", "72 bare, 36 interleaved 1 MFMA : 3 VALU", "72 bare, 36 interleaved 1 : 4",
                        "72 bare, 36 interleaved 1 : 5", "72 bare, 36 interleaved 1 : 2", "108 bare + cell update, no early half (ref)",
                        "72 bare, 36 early, by hand: slice per MFMA", "72 bare, 36 early, slice per 2nd MFMA", "72 bare, 48 early, slice per 2nd MFMA (120 MFMAs)", "72 bare, 24 early, slice per MFMA (96 MFMAs)"};
    for (int mode = 0; mode < 10; ++mode) {
        run(Args{2, mode, 0, 8, 400}, h);
        printf("tick %-46s: %7.0f cycles per tick (216 MFMAs per SIMD = 6912 of pipe time)\n", tn[mode], (double)h[0] / 400);
    }
    return 0;
}
