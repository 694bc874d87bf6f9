// aidax_device.h — device-side helpers shared by the gfx950 kernels: cross-lane DPP/permlane ops,
// the activation functions, the two value smoothers, the systolic biquad pass and block I/O.
// See aidax_kernels.hip for how they are used.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "aidax_layout.h"

namespace aidax {

// ------------------------------------------------------------------ lane ops
__device__ __forceinline__ float dpp_row_shr1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf));
}

// Sum over the 64 lanes, returned wave-uniform (tree order fixed by the DPP network).
__device__ __forceinline__ float wave_sum(float v)
{
    v = v + dpp_take<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = v + dpp_take<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = v + dpp_take<0x141, 0xf>(v);    // row_half_mirror
    v = v + dpp_take<0x140, 0xf>(v);    // row_mirror: every lane of a row holds the row sum
    v = v + dpp_take<0x142, 0xa>(v);    // row_bcast:15 into rows 1,3
    v = v + dpp_take<0x143, 0xc>(v);    // row_bcast:31 into rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// acc += w * (h of the lane N places to the left in its 16-lane row, cyclically): the cross-lane read rides on the FMA
// itself (v_fmac_f32_dpp). The compiler's DPP combine does not fold v_mov_dpp into a following fmac here, so the
// instruction is spelled out. The hazard recognizer does not look into inline asm: `h` must have been written at least
// two instructions before the first rotation (VALU write -> DPP read needs two wait states) — in LstmCell it is the
// previous frame's result, with the publish, the LDS reads and the input FMAs in between; an explicit s_nop costs an
// issue slot of the lone wave per use (measured: cfg2 73.4 -> 75.4 us) and is not needed there.
template <int N>
__device__ __forceinline__ void fmac_row_ror(float& acc, float h, float w)
{
    static_assert(N >= 1 && N <= 15, "rotation within a row of 16 lanes");
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w), "n"(N));
}

// Rotations 1..15 of ONE accumulation chain in ONE asm statement. Issued one by one, the compiler's hazard recognizer cannot
// see that the accumulator an asm statement reads is not its DPP operand and puts an s_nop between every pair of them — 14
// issue slots of the lone recurrent wave per LSTM-32 frame: cfg2 73.5 -> 69.5 us without them. Inside one statement nothing
// is inserted; the real rule (a VGPR written by a VALU instruction needs two wait states before a DPP instruction reads it AS
// ITS DPP OPERAND) concerns `h` only, see fmac_row_ror above. A cell with two gate rows per lane (LSTM-32) issues one
// statement per row: the second reads nothing the first wrote, so nothing is put between them either. Same instructions
// per chain in the same order as fmac_row_ror<1..15>.
#define AIDAX_ROR1(N, W) "v_fmac_f32_dpp %0, %1, %" #W " row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void fmac_row_ror_1_15(float& a0, float h, const float* w)
{
    asm volatile(AIDAX_ROR1(1, 2) AIDAX_ROR1(2, 3) AIDAX_ROR1(3, 4) AIDAX_ROR1(4, 5) AIDAX_ROR1(5, 6) AIDAX_ROR1(6, 7) AIDAX_ROR1(7, 8)
                 AIDAX_ROR1(8, 9) AIDAX_ROR1(9, 10) AIDAX_ROR1(10, 11) AIDAX_ROR1(11, 12) AIDAX_ROR1(12, 13) AIDAX_ROR1(13, 14)
                 AIDAX_ROR1(14, 15) AIDAX_ROR1(15, 16)
                 : "+v"(a0)
                 : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]),
                   "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
}
#undef AIDAX_ROR1

// Cross-lane shares on the permlane swap network. (The swapped pair is copied to
// scalars before the float bit_cast: __builtin_bit_cast applied directly to an
// ext-vector element reads element 0 on ROCm 7.2's clang.)
struct Pair { float lo, hi; };

// every lane gets (value held by its lane in the low half, value held in the high half)
__device__ __forceinline__ Pair share_halves(float v)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return { __builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1) };
}

// every lane gets (value of the even 16-lane row of its row pair, value of the odd row)
__device__ __forceinline__ Pair share_rows(float v)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return { __builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1) };
}

__device__ __forceinline__ float fast_sigmoid(float v)
{
    // 1 / (1 + 2^(-v*log2 e)); saturates cleanly: exp2 -> inf gives rcp -> 0
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896340736f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// The same for a pre-activation that already carries the factor -log2(e): the table kernels' packers fold it into the
// sigmoid rows' weights and biases (aidax_pack.cpp), which takes the multiply off the recurrence's critical path.
constexpr float kNegLog2e = -1.44269504088896340736f;
__device__ __forceinline__ float sigmoid_pre(float v_scaled)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v_scaled));
}

// tanh as an odd rational x*P(x^2)/Q(x^2), P of degree 6 and Q of degree 3 in x^2, fitted for
// this kernel (scratch: Lawson-weighted least squares on [0,9], relative error 6.6e-9 in fp64).
// In fp32 with FMAs: <= 3.6e-7 relative everywhere, exact odd symmetry, |y| <= 1 with the
// clamp. The relative accuracy near 0 is the point: tanh = 2*sigmoid(2x)-1 on v_exp_f32 is
// only ABSOLUTELY accurate (1.2e-7), and an LSTM whose forget gate sits near 1 integrates
// that error in c (measured 5.5e-6 after 2048 samples on tw40_british_lead vs 1.3e-6 with
// this form; the reference's threshold is 1e-5, rt-neural-generic.h:182).
// (Round 3 looked for a shorter one — scratch/fit_tanh3.py, fit_tanh4.py: degree 4 over degree 4 has the same 5-ulp bound
// with one FMA less, x + x*u*R/Q is 1 ulp below |x| = 1 but cancels above 2 — and kept this one: every candidate's rounding
// error is a deterministic function of x, a cell at rest evaluates the same x every frame, and the 2048-frame warm-up of
// tw40_british_lead ended 4.7e-6 from the oracle's state with the shorter rational against 1.6e-6 with this one, for 1.3 % of
// cfg2's time.)
constexpr float kTanhP[7] = { 0.9999999933888696f, 0.13084010352004496f, 0.003103956503888039f, 1.1154311501654368e-05f,
                              -2.0225239996482085e-08f, 5.277955823366522e-11f, -8.488730763828322e-14f };      // P, ascending in x^2
constexpr float kTanhQ[4] = { 1.0f, 0.46417337453820245f, 0.02449517952619233f, 0.00025461456545097517f };     // Q, ascending in x^2
__device__ __forceinline__ float tanh_rat_clamp(float v) { return __builtin_fminf(__builtin_fmaxf(v, -7.9f), 7.9f); }     // (v_max + v_med3; a bare v_med3 lets a NaN through and is no faster)
__device__ __forceinline__ float tanh_rat(float v)
{
    const float x = tanh_rat_clamp(v);
    const float u = x * x;
#ifndef AIDAX_TANH_PLAIN
    // The first three Horner steps of P and all three of Q side by side in v_pk_fma_f32 ({p, q} <- {p, q} * {u, u} + {cP, cQ}: the same FMAs on the same
    // operands, so the same bits), the last three of P alone: six instructions for nine. A lone wave issues a packed FMA in a plain one's interval
    // (profiles/r05_lone_wave_issue.txt), and the recurrent wave's frame is its instruction count (profiles/r05_cfg2_frame_trace.txt): two rationals per frame.
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t uu = { u, u };
    f32x2_t pq = { kTanhP[6], kTanhQ[3] };
    pq = __builtin_elementwise_fma(pq, uu, f32x2_t{ kTanhP[5], kTanhQ[2] });
    pq = __builtin_elementwise_fma(pq, uu, f32x2_t{ kTanhP[4], kTanhQ[1] });
    pq = __builtin_elementwise_fma(pq, uu, f32x2_t{ kTanhP[3], kTanhQ[0] });
    float pp = pq.x;
    pp = __builtin_fmaf(pp, u, kTanhP[2]);
    pp = __builtin_fmaf(pp, u, kTanhP[1]);
    pp = __builtin_fmaf(pp, u, kTanhP[0]);
    return (pp * x) * __builtin_amdgcn_rcpf(pq.y);
#endif
    float p = kTanhP[6];
    p = __builtin_fmaf(p, u, kTanhP[5]);
    p = __builtin_fmaf(p, u, kTanhP[4]);
    p = __builtin_fmaf(p, u, kTanhP[3]);
    p = __builtin_fmaf(p, u, kTanhP[2]);
    p = __builtin_fmaf(p, u, kTanhP[1]);
    p = __builtin_fmaf(p, u, kTanhP[0]);
    float q = kTanhQ[3];
    q = __builtin_fmaf(q, u, kTanhQ[2]);
    q = __builtin_fmaf(q, u, kTanhQ[1]);
    q = __builtin_fmaf(q, u, kTanhQ[0]);
    return (p * x) * __builtin_amdgcn_rcpf(q);
}

// tanh for FEED-FORWARD layers (the conv1d stacks): 1 - 2 / (1 + e^(2x)) on v_exp_f32 / v_rcp_f32, five instructions
// against fifteen for the rational. Absolutely accurate to ~1.2e-7, not relatively near 0 — which only matters where
// a state integrates the error (the LSTM cell above); a conv stack has no such state, eight layers measured
// 1.9e-7 against the oracle (tests/test_gpu_parity.py). Saturates cleanly: e -> inf gives 1, e -> 0 gives -1.
__device__ __forceinline__ float tanh_exp(float v)
{
    const float e = __builtin_amdgcn_exp2f(v * 2.88539008177792681472f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// ... for a pre-activation that already carries the factor 2 log2(e) (the conv packer folds it into tanh layers' weights
// and biases): four instructions
__device__ __forceinline__ float tanh_exp_pre(float v_scaled)
{
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v_scaled)), 1.0f);
}

// ---------------------------------------------------------------- smoothers
struct ExpRamp {              // ExponentialValueSmoother::next, ValueSmoother.hpp:142-145
    float mem, coef, tc;      // tc = target * (1.f - coef), loop-invariant
    __device__ __forceinline__ void arm(float m, float target, float c)
    {
        mem = m; coef = c; tc = target * (1.f - c);
    }
    __device__ __forceinline__ float next() { mem = mem * coef + tc; return mem; }
};

__device__ __forceinline__ float lin_next(float& mem, float target, float step)
{
    // LinearValueSmoother::next, ValueSmoother.hpp:229-234
    const float y0 = mem;
    const float dy = target - y0;
    mem = y0 + __builtin_copysignf(__builtin_fminf(__builtin_fabsf(dy), __builtin_fabsf(step)), dy);
    return mem;
}

// ------------------------------------------------------------ systolic chain
// One pass of up to 6 cascaded stages over buf[0..n) in LDS, in place.
// Lane k < K owns stage k: an optional biquad (slot[k], enabled by act[k]) and,
// on lane `gain_lane`, the exponential gain ramp applied after the biquad.
struct ChainPass {
    int K, gain_lane;
    bool active;          // this lane's biquad is in circuit
    double a0, a1, a2, b1, b2, z1, z2;
    ExpRamp g;
};

// One systolic step of one lane. CHECK = false is the steady state (every stage of the cascade has a
// sample in range): no range tests, no exec-mask branches, the biquad is evaluated unconditionally and a
// bypassed stage just selects x — ~27 instructions instead of ~45 with three nested branches. A lone
// wave issues one instruction per ~5 cycles, so this loop IS the cost of a chain pass. Lanes that do not
// run (stage >= K, disabled stream) and bypassed stages compute on whatever flows by; the caller restores
// their z1/z2 afterwards (chain_finish).
template <int DST_STRIDE, bool CHECK>
__device__ __forceinline__ void chain_step(ChainPass& c, int stage, bool run, bool writer, float head, float& carry,
                                           float* dst, int s, int n)
{
    const float from_left = dpp_row_shr1(carry);
    const float x = stage == 0 ? head : from_left;
    const int idx = s - stage;
    if (!CHECK || (run && idx >= 0 && idx < n)) {
        const double xd = x;                                // Biquad::process, Biquad.h:53-58
        const double yd = xd * c.a0 + c.z1;
        c.z1 = xd * c.a1 + c.z2 - c.b1 * yd;
        c.z2 = xd * c.a2 - c.b2 * yd;
        float y = c.active ? (float)yd : x;
        const float gm = c.g.next();                        // every stage lane keeps its own copy;
        if (stage == c.gain_lane) y = y * gm;               // only the gain lane's is used
        carry = y;
        if (writer) dst[idx * DST_STRIDE] = y;
    }
}

// Runs the pass over src[0..n) -> dst. `stage` is this lane's position in its cascade (the lane id, or
// lane & 7 when one wave carries eight streams), `run` whether the lane owns a stage of a running stream,
// `depth` the longest cascade in the wave (steps = n + depth - 1).
template <int DST_STRIDE = 1>
__device__ __forceinline__ void chain_sweep(ChainPass& c, int stage, bool run, int depth, const float* src, float* dst, int n)
{
    const double z1o = c.z1, z2o = c.z2;
    const bool writer = run && stage == c.K - 1;
    float carry = 0.f;                         // this lane's previous output, read by lane+1
    const int steps = n + depth - 1;
    const int fill = depth - 1 < steps ? depth - 1 : steps;
    float head_next = src[0];                  // LDS broadcast, fetched one step ahead
    int s = 0;
    for (; s < fill; ++s) {
        const float head = head_next;
        head_next = src[s + 1 < n ? s + 1 : n - 1];
        chain_step<DST_STRIDE, true>(c, stage, run, writer, head, carry, dst, s, n);
        __builtin_amdgcn_wave_barrier();
    }
    for (; s < n; ++s) {                       // steady state: all `depth` stages in range
        const float head = head_next;
        head_next = src[s + 1 < n ? s + 1 : n - 1];
        chain_step<DST_STRIDE, false>(c, stage, run, writer, head, carry, dst, s, n);
        __builtin_amdgcn_wave_barrier();
    }
    for (; s < steps; ++s) {
        chain_step<DST_STRIDE, true>(c, stage, run, writer, head_next, carry, dst, s, n);
        __builtin_amdgcn_wave_barrier();
    }
    if (!run || !c.active) { c.z1 = z1o; c.z2 = z2o; }     // a bypassed biquad keeps its state (:622, :646)
}

// The same pass over a whole block, BLOCKED in time: a stage takes kChainBlock frames per hand-over
// instead of one, so the per-step overhead of the systolic form (lane shift, selects, range logic, loop)
// is paid once per eight samples and the eight biquad evaluations of a lane run back to back. Stage k works
// on frame block m-k at macro-step m; blocks travel to the next lane through a double-buffered LDS slot
// (`hand`: [2][64 lanes][kChainBlock] floats). Same operations in the same order per sample and per stage as
// chain_step, so the results are bit-identical. `row` is processed in place; n_full is a multiple of
// kChainBlock (the caller finishes a ragged tail with chain_sweep).
#ifndef AIDAX_CHAIN_BLOCK
#define AIDAX_CHAIN_BLOCK 8
#endif
constexpr int kChainBlock = AIDAX_CHAIN_BLOCK;

// PLAIN: every lane that runs has its biquad in circuit and every gain ramp of the wave sits on its fixed point
// (mem * coef + tc == mem, so next() returns the same value for the rest of the block): the per-sample select
// and the two ramp instructions drop out — 12 instructions per sample and stage instead of 15, same values.
// B frames per hand-over (kChainBlock, or more where the hand-over dominates: the fused conv kernel's lone chain wave), HL the
// number of lanes that have a slot in `hand` ([2][HL][B] floats).
template <bool PLAIN, int B = kChainBlock, int HL = kWave>
__device__ __forceinline__ void chain_macro_step(ChainPass& c, ExpRamp& g, int stage, bool run, bool last,
                                                 float* row, float* hand, int M, int m, int lane)
{
    const int j = m - stage;
    if (run && j >= 0 && j < M) {
        const float g_fixed = g.mem;
        const float* src = stage == 0 ? row + B * j : hand + (((m - 1) & 1) * HL + lane - 1) * B;
        float* dst = last ? row + B * j : hand + ((m & 1) * HL + lane) * B;
        float v[B];
#pragma unroll
        for (int q = 0; q < B / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const float x = v[i];
            const double xd = x;                            // Biquad::process, Biquad.h:53-58
            const double yd = xd * c.a0 + c.z1;
            c.z1 = xd * c.a1 + c.z2 - c.b1 * yd;
            c.z2 = xd * c.a2 - c.b2 * yd;
            if constexpr (PLAIN) {
                v[i] = (float)yd * g_fixed;
            } else {
                const float y = c.active ? (float)yd : x;
                v[i] = y * g.next();
            }
        }
#pragma unroll
        for (int q = 0; q < B / 4; ++q)
            *reinterpret_cast<float4*>(dst + 4 * q) = float4{ v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3] };
    }
    __builtin_amdgcn_wave_barrier();
}

template <bool PLAIN, int B = kChainBlock, int HL = kWave>
__device__ __forceinline__ void chain_blocks(ChainPass& c, ExpRamp& g, int stage, bool run, int depth, bool last,
                                             float* row, float* hand, int M, int lane)
{
    for (int m = 0; m < M + depth - 1; ++m) chain_macro_step<PLAIN, B, HL>(c, g, stage, run, last, row, hand, M, m, lane);
}

template <int B = kChainBlock, int HL = kWave>
__device__ __forceinline__ void chain_sweep_blocked(ChainPass& c, int stage, bool run, int depth, float* row, float* hand,
                                                    int n_full, int lane, bool force_general = false)
{
    const double z1o = c.z1, z2o = c.z2;
    const int M = n_full / B;
    ExpRamp g = c.g;
    const bool is_gain = stage == c.gain_lane;
    if (!is_gain) { g.mem = 1.f; g.coef = 1.f; g.tc = 0.f; }     // y * 1.0f is exact: no select per sample
    const bool last = stage == c.K - 1;
    // wave-uniform: does any running lane need the select (a bypassed biquad) or a moving ramp?
    const bool fussy = run && (!c.active || g.mem * g.coef + g.tc != g.mem);
    if (!force_general && __builtin_amdgcn_ballot_w64(fussy) == 0) chain_blocks<true, B, HL>(c, g, stage, run, depth, last, row, hand, M, lane);
    else chain_blocks<false, B, HL>(c, g, stage, run, depth, last, row, hand, M, lane);
    if (is_gain) c.g = g;
    if (!run || !c.active) { c.z1 = z1o; c.z2 = z2o; }
}

// The blocked pass as a RESUMABLE job: begin, one macro-step (eight frames per stage) at a time, end — for a wave that
// has other duties between the steps (the helper waves of k_mfma_lp's one-launch form run the pre pass ahead of the
// frame loop and the post pass behind it, a step per tick). Same operations per sample and stage as chain_sweep_blocked.
struct ChainJob {
    ExpRamp g;
    double z1o, z2o;
    int M, m;
    bool last, is_gain, plain;
};
__device__ __forceinline__ void chain_job_begin(ChainJob& r, const ChainPass& c, int stage, bool run, int n_full, bool force_general)
{
    r.z1o = c.z1; r.z2o = c.z2;
    r.M = n_full / kChainBlock;
    r.m = 0;
    r.g = c.g;
    r.is_gain = stage == c.gain_lane;
    if (!r.is_gain) { r.g.mem = 1.f; r.g.coef = 1.f; r.g.tc = 0.f; }
    r.last = stage == c.K - 1;
    const bool fussy = run && (!c.active || r.g.mem * r.g.coef + r.g.tc != r.g.mem);
    r.plain = !force_general && __builtin_amdgcn_ballot_w64(fussy) == 0;
}
__device__ __forceinline__ bool chain_job_pending(const ChainJob& r, int depth) { return r.M != 0 && r.m < r.M + depth - 1; }
// frames the FIRST stage needs in the row for the next step (an input that arrives over time: the post pass)
__device__ __forceinline__ int chain_job_needs(const ChainJob& r) { return r.m < r.M ? kChainBlock * (r.m + 1) : 0; }
__device__ __forceinline__ void chain_job_step(ChainJob& r, ChainPass& c, int stage, bool run, float* row, float* hand, int lane)
{
    if (r.plain) chain_macro_step<true>(c, r.g, stage, run, r.last, row, hand, r.M, r.m, lane);
    else chain_macro_step<false>(c, r.g, stage, run, r.last, row, hand, r.M, r.m, lane);
    ++r.m;
}
__device__ __forceinline__ void chain_job_end(const ChainJob& r, ChainPass& c, bool run)
{
    if (r.is_gain) c.g = r.g;
    if (!run || !c.active) { c.z1 = r.z1o; c.z2 = r.z2o; }
}

template <int DST_STRIDE = 1>
__device__ __forceinline__ void chain_run(ChainPass& c, const float* src, float* dst, int n, int lane)
{
    chain_sweep<DST_STRIDE>(c, lane, lane < c.K, c.K, src, dst, n);
}

// One stream's pass by one wave (lane k = stage k), in place, in the blocked form: whole blocks of frames through `hand`
// (kChainHandFloats floats of LDS), a ragged tail sample by sample. Same values as chain_run.
// This wave — the lone chain wave of the fused conv kernel — hands SIXTEEN frames over at a time: a macro-step's fixed part (the LDS
// turn-around between two stages, range logic: ~230 of ~680 cycles at eight frames, scratch/conv_trace.py) is paid half as
// often, which outweighs the five steps a six-stage cascade takes to fill being twice as long (cfg4: pre pass 9.2 -> 7.9 us,
// post pass 10.5 -> 10.1 us). Its cascades are at most six stages deep: eight lanes' worth of slots, [2][8][16] floats.
constexpr int kChainHandFloats = 2 * kWave * kChainBlock;
// (Thirty-two frames per step for a one-stage pass, which hands nothing over, measured no better: 7.95 against 7.88 us, and
// uneven from CU to CU.)
constexpr int kChainWideBlock = 16, kChainWideLanes = 8;
static_assert(2 * kChainWideLanes * kChainWideBlock <= kChainHandFloats, "the wide form lives in the same hand-over area");
__device__ __forceinline__ void chain_run_blocked(ChainPass& c, float* buf, float* hand, int n, int lane)
{
    const bool run = lane < c.K;
    const int n_full = n & ~(kChainWideBlock - 1);
    if (n_full != 0) chain_sweep_blocked<kChainWideBlock, kChainWideLanes>(c, lane, run, c.K, buf, hand, n_full, lane);
    if (n_full != n) chain_sweep<1>(c, lane, run, c.K, buf + n_full, buf + n_full, n - n_full);
}


// ------------------------------------------------------------- block I/O
__device__ __forceinline__ void load_block(float* buf, const float* __restrict__ src, int n, int lane)
{
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* b4 = reinterpret_cast<float4*>(buf);
        for (int i = lane; i < n / 4; i += kWave) b4[i] = s4[i];     // 16 B/lane, 1 KiB per wave instruction
    } else {
        for (int i = lane; i < n; i += kWave) buf[i] = src[i];
    }
}

__device__ __forceinline__ void store_block(float* __restrict__ dst, const float* buf, int n, int lane)
{
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        float4* d4 = reinterpret_cast<float4*>(dst);
        const float4* b4 = reinterpret_cast<const float4*>(buf);
        for (int i = lane; i < n / 4; i += kWave) d4[i] = b4[i];
    } else {
        for (int i = lane; i < n; i += kWave) dst[i] = buf[i];
    }
}

// Stage -> biquad slot of the two systolic passes (cascade order of
// applyToneControls, rt-neural-generic.cpp:133-139)
__device__ __forceinline__ int pre_slot(int k)  { return k == 0 ? BQ_LPF : BQ_DEPTH + (k - 1); }
__device__ __forceinline__ int post_slot(int k) { return k == 0 ? BQ_DC : BQ_DEPTH + (k - 1); }

__device__ __forceinline__ void chain_load(ChainPass& c, const StreamCtl& ctl, const StreamState& st, int slot, bool act)
{
    c.active = act;
    c.a0 = ctl.bq[slot][0]; c.a1 = ctl.bq[slot][1]; c.a2 = ctl.bq[slot][2];
    c.b1 = ctl.bq[slot][3]; c.b2 = ctl.bq[slot][4];
    c.z1 = st.z[slot][0];   c.z2 = st.z[slot][1];
}

// Measurement build only (make EXTRA=-DAIDAX_CONV_TRACE ..., scratch/conv_trace.py): shader-clock stamps of the fused conv
// kernel's chain wave, left in the first floats of the stream's output row.
#ifdef AIDAX_CONV_TRACE
__device__ __forceinline__ unsigned long long* cv_trace() { __shared__ unsigned long long t[32]; return t; }
#define CV_STAMP(k) do { if ((threadIdx.x & 63) == 0) cv_trace()[k] = clock64(); } while (0)
#else
#define CV_STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------ shared chain pieces
// The per-stream prologue + pre pass of run() (:489-518, :607-630) executed by ONE wave on
// an LDS block buffer; ChainCtx::live is false when the stream early-outs (pre-run / disabled).
struct ChainCtx {
    uint32_t flags, pending;
    float pre_mem, master_mem, pre_tgt, master_tgt;
    bool live;
};

// EAGER (the fused conv kernel, where this is the head of the launch's critical path): the audio row and the coefficients
// and state of every stage a pre pass can have are requested before the control word they depend on is waited for — one
// trip to memory instead of three in a row. A lane beyond the cascade then holds an EQ stage's numbers instead of stage 0's;
// it does not run either way.
template <bool EAGER = false>
__device__ __forceinline__ ChainCtx chain_prologue(const StreamCtl& ctl, StreamState& st, const float* in_row,
                                                   float* out_row, float* buf, int n, int lane, float* hand = nullptr)
{
    ChainCtx c;
    // (a time slice of a longer block keeps the block's pitch: rows start 16-byte aligned only if that pitch is a multiple of four frames)
    const bool row_in_regs = EAGER && (n & 3) == 0 && n <= 4 * kWave && (reinterpret_cast<uintptr_t>(in_row) & 15u) == 0;
    float4 rowv = float4{ 0.f, 0.f, 0.f, 0.f };
    ChainPass p;
    if constexpr (EAGER) {
        if (row_in_regs && 4 * lane < n) rowv = reinterpret_cast<const float4*>(in_row)[lane];
        chain_load(p, ctl, st, pre_slot(lane < 6 ? lane : 0), false);
    }
    c.flags = ctl.flags;
    c.pending = st.pending;
    c.pre_mem = st.pre_mem; c.master_mem = st.master_mem;
    c.pre_tgt = st.pre_tgt; c.master_tgt = st.master_tgt;
    if (c.pending & PEND_ACTIVATE) {
        c.pre_mem = c.pre_tgt;
        c.master_mem = c.master_tgt;
        c.pending &= ~PEND_ACTIVATE;
    }
    c.pre_tgt = ctl.pre_target;
    c.live = !(n == 0 || !(c.flags & CTL_ENABLED));
    if (!c.live) {
        if (n != 0 && out_row != in_row)
            for (int i = lane; i < n; i += kWave) out_row[i] = in_row[i];
        if (lane == 0) { st.pre_mem = c.pre_mem; st.master_mem = c.master_mem; st.pre_tgt = c.pre_tgt; st.pending = c.pending; }
        return c;
    }
    CV_STAMP(1);
    if (row_in_regs) { if (4 * lane < n) reinterpret_cast<float4*>(buf)[lane] = rowv; }
    else load_block(buf, in_row, n, lane);
    __builtin_amdgcn_wave_barrier();
    const bool eq = c.flags & CTL_EQ_PRE;
    p.K = eq ? 6 : 1;
    p.gain_lane = 0;
    const int k = lane < p.K ? lane : 0;
    const int slot = pre_slot(k);
    const bool act = k == 0 ? (c.flags & CTL_LPF_ON) != 0 : ((c.flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
    if constexpr (EAGER) p.active = act;
    else chain_load(p, ctl, st, slot, act);
    p.g.arm(c.pre_mem, c.pre_tgt, ctl.pre_coef);
#ifdef AIDAX_CONV_TRACE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    CV_STAMP(2);
    if (hand) chain_run_blocked(p, buf, hand, n, lane);       // kChainHandFloats of LDS: the blocked form
    else chain_run(p, buf, buf, n, lane);
    CV_STAMP(3);
    if (lane < p.K) { st.z[slot][0] = p.z1; st.z[slot][1] = p.z2; }
    c.pre_mem = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p.g.mem), 0));
    return c;
}

// One of the two packed passes of run() by ONE wave over EIGHT streams (8 lanes per stream, lane & 7 = cascade stage) whose
// rows already sit in LDS (rows + slot * nP), in place: PRE = LPF -> pre-gain ramp -> EQ(pre), otherwise DC blocker ->
// EQ(post) -> master ramp. The body of k_chain; k_mfma_lp's stacked form runs it on its first and last layer's workgroups.
// `commit`: write the biquad states and the ramp's memory / target back (a workgroup that only needs the pass's OUTPUT —
// the last layer re-deriving the model input — passes false). `atomic_pending`: clear PEND_ACTIVATE with an atomic (another
// workgroup of the same launch may be clearing PEND_PARAM_FIRST in the same word).
constexpr int kChainWaveStreams = 8;
// Sixteen frames per hand-over here too (see chain_run_blocked: a macro-step's fixed part is a third of it at eight): a packed
// pass of 256 frames 10.6 -> ~9 us of loop. The wave's hand-over area is [2][64 lanes][16] floats.
constexpr int kChainPackBlock = 16;
constexpr int kChainPackHandFloats = 2 * kWave * kChainPackBlock;
template <bool PRE>
__device__ __forceinline__ void chain_wave_pass(const LaunchArgs& a, int stream0, float* rows, int nP, float* hand, int n, int lane,
                                                bool commit, bool atomic_pending)
{
    const int grp = lane >> 3, stage = lane & 7;
    const int sg = stream0 + grp;
    const bool valid = sg < (int)a.n_streams;
    const int sc = valid ? sg : (int)a.n_streams - 1;          // clamp: invalid groups shadow the last stream, never store
    const StreamCtl& ctl = a.ctl[sc];
    StreamState& st = a.st[sc];
    const uint32_t flags = ctl.flags;
    const uint32_t pending0 = st.pending;
    const bool live = valid && n != 0 && (flags & CTL_ENABLED);

    ChainPass c;
    const bool eq = flags & (PRE ? CTL_EQ_PRE : CTL_EQ_POST);
    c.K = eq ? 6 : 1;
    c.gain_lane = PRE ? 0 : c.K - 1;
    const int k = stage < c.K ? stage : 0;
    const int slot = PRE ? pre_slot(k) : post_slot(k);
    const bool act = k == 0 ? (flags & (PRE ? CTL_LPF_ON : CTL_DC_ON)) != 0
                            : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
    chain_load(c, ctl, st, slot, act);
    float mem = PRE ? st.pre_mem : st.master_mem;
    float tgt = PRE ? st.pre_tgt : st.master_tgt;
    if (pending0 & PEND_ACTIVATE) mem = tgt;                   // activate(): clearToTargetValue (:341-342)
    if (PRE) tgt = ctl.pre_target;                             // :513, before every early-out
    else if (live) tgt = ctl.master_target;                    // :654, only on the DSP path
    c.g.arm(mem, tgt, PRE ? ctl.pre_coef : ctl.master_coef);

    if (n != 0) {
        // whole blocks of sixteen frames in the blocked form, a ragged tail sample by sample
        float* row = rows + grp * nP;
        const bool run = live && stage < c.K;
        const int n_full = n & ~(kChainPackBlock - 1);
        // the longest cascade among the wave's running streams: without EQ on this side it is one stage, and the
        // sweep needs no fill / drain steps at all
        const int depth = (AIDAX_TUNE(a) & 8) || __builtin_amdgcn_ballot_w64(run && stage > 0) != 0 ? 6 : 1;
        if (n_full != 0) chain_sweep_blocked<kChainPackBlock, kWave>(c, stage, run, depth, row, hand, n_full, lane, (AIDAX_TUNE(a) & 8) != 0);
        if (n_full != n) chain_sweep<1>(c, stage, run, depth, row + n_full, row + n_full, n - n_full);
        __builtin_amdgcn_wave_barrier();
    }
    if (!valid || !commit) return;
    if (live && stage < c.K) { st.z[slot][0] = c.z1; st.z[slot][1] = c.z2; }
    if (stage == c.gain_lane) {
        const float m_out = live ? c.g.mem : mem;
        if (PRE) { st.pre_mem = m_out; st.pre_tgt = tgt; }
        else {
            st.master_mem = m_out; st.master_tgt = tgt;
            if (atomic_pending) __hip_atomic_fetch_and(&st.pending, ~(uint32_t)PEND_ACTIVATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else st.pending = st.pending & ~PEND_ACTIVATE;
        }
    }
}

// run() :634-640 for a model that takes no PARAM input (the conv stacks): the smoothers' targets still follow the
// controls and the first run after a load still snaps them — what a model swapped in later inherits (:822-825).
// Nothing calls next(), so the memories move only on that snap. One lane; returns `pending` with the flag cleared.
__device__ __forceinline__ uint32_t param_targets(const StreamCtl& ctl, StreamState& st, uint32_t pending)
{
    float p_mem[2] = { st.p_mem[0], st.p_mem[1] };
#pragma unroll
    for (int i = 0; i < 2; ++i) {                        // LinearValueSmoother::setTargetValue (:209-216)
        const float nt = ctl.p_target[i];
        if (__builtin_fabsf(st.p_tgt[i] - nt) >= FLT_EPSILON) {
            st.p_tgt[i] = nt;
            st.p_step[i] = (nt - p_mem[i]) / ctl.p_den;
        }
    }
    if (pending & PEND_PARAM_FIRST) {                    // paramFirstRun (:636-640)
        pending &= ~PEND_PARAM_FIRST;
        st.p_mem[0] = st.p_tgt[0];
        st.p_mem[1] = st.p_tgt[1];
    }
    return pending;
}

// post pass + store + state write-back (:645-655), in two halves: _begin requests what the pass needs from memory (a caller
// with work left for the wave before the pass — the fused conv kernel's Dense — puts it in between), _run is the pass.
__device__ __forceinline__ ChainPass chain_epilogue_begin(const StreamCtl& ctl, const StreamState& st, ChainCtx& c, int lane)
{
    c.master_tgt = ctl.master_target;
    ChainPass p;
    const bool eq = c.flags & CTL_EQ_POST;
    p.K = eq ? 6 : 1;
    p.gain_lane = p.K - 1;
    const int k = lane < p.K ? lane : 0;
    const int slot = post_slot(k);
    const bool act = k == 0 ? (c.flags & CTL_DC_ON) != 0 : ((c.flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
    chain_load(p, ctl, st, slot, act);
    p.g.arm(c.master_mem, c.master_tgt, ctl.master_coef);
    return p;
}
__device__ __forceinline__ void chain_epilogue_run(StreamState& st, ChainCtx& c, ChainPass& p, float* out_row,
                                                   float* buf, int n, int lane, float* hand = nullptr)
{
    const int slot = post_slot(lane < p.K ? lane : 0);
#ifdef AIDAX_CONV_TRACE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    CV_STAMP(10);
    if (hand) chain_run_blocked(p, buf, hand, n, lane);
    else chain_run(p, buf, buf, n, lane);
    CV_STAMP(11);
    if (lane < p.K) { st.z[slot][0] = p.z1; st.z[slot][1] = p.z2; }
    c.master_mem = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p.g.mem), p.K - 1));
    __builtin_amdgcn_wave_barrier();
    store_block(out_row, buf, n, lane);
    if (lane == 0) {
        st.pre_mem = c.pre_mem; st.master_mem = c.master_mem;
        st.pre_tgt = c.pre_tgt; st.master_tgt = c.master_tgt;
        st.pending = c.pending;
    }
}
__device__ __forceinline__ void chain_epilogue(const StreamCtl& ctl, StreamState& st, ChainCtx& c, float* out_row,
                                               float* buf, int n, int lane, float* hand = nullptr)
{
    ChainPass p = chain_epilogue_begin(ctl, st, c, lane);
    chain_epilogue_run(st, c, p, out_row, buf, n, lane, hand);
}

}  // namespace aidax
