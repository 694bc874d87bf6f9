// aidax_mfmalp.hip — k_mfma_lp: STACKED recurrent models on the matrix cores, one workgroup per (16 streams, LAYER),
// the layers of a stream group chained through a small ring in global memory.
//
// Why: k_mfma (aidax_mfma.hip) gives a group of 16 streams ONE workgroup that walks all layers, streaming 437 KiB of
// A fragments from L2 every frame (LSTM-96 x2). At BASELINE cfg5's per-GPU size (2048 streams = 128 groups) that
// fills half of the 256 CUs, and the fragment stream plus its address arithmetic sits between the MFMAs. Here
//   * a group's layers run on SEPARATE workgroups (2048 streams x 2 layers = 256 workgroups: every CU busy), layer l
//     one or more frames behind layer l-1;
//   * each workgroup keeps ITS layer's A fragments in registers for the whole launch (LSTM-96 layer 1: 384 rows x
//     192 columns = 288 KiB = 144 registers per lane over 8 waves): the frame loop is MFMAs fed by one
//     conflict-free ds_read_b32 per k-step, no global loads, no address arithmetic;
//   * h of layer l-1 travels to layer l through a ring of kLpRing frames in global memory, in the layout layer l
//     reads it in ([unit][stream] = ready B fragments). ONE-directional hand-over in the write-through form of
//     cdna_hip_programming.md G16: the payload leaves as 16-byte device-scope (sc1) stores and is read with 16-byte
//     sc1 loads, so no L2 write-back and no L1 invalidate is ever needed (measured here: an agent-scope release costs
//     the publishing wave ~18 k cycles, 16 producers per XCD flushing one L2); the producer publishes a frame counter
//     every kLpBatch frames after its stores have drained (per-wave s_waitcnt vmcnt(0) -> barrier), the consumer's lane
//     0 polls it only when it has used up what it knows of; the producer waits only when the ring is full.
//     Latency of the hand-over is hidden by the skew between the layers, nobody waits per frame.
//
// Forward progress: the pool uses this kernel only while groups x layers <= the number of CUs (one workgroup
// fits per CU), so EVERY workgroup of the grid becomes resident whatever the dispatch order, and a spinning consumer
// can never keep its producer off the machine. (Forced on larger pools with AIDAX_MFMA_LP=1 — measurements only —
// it leans on the dispatcher's observed id order: a layer's workgroup has a LOWER id than the layer above it, ids of
// one group are 8 apart.) Every spin is bounded in time. The premise also needs the grid to have the CUs to itself: the
// pool lets ONE pool per device use this kernel, warms models up with k_mfma and never puts two of these grids in flight
// (aidax_pool.cpp, lp_gate); what it cannot see (another process on the GPU) ends in a reported give-up. Nothing depends on placement: ids of one group being equal modulo
// 8 puts them on one XCD under the observed b % 8 placement, a locality hint only.
//
// Same arithmetic as k_mfma (same fragments from pack_mfma, same accumulation order, same activations): the two
// kernels leave bit-identical recurrent state on the same model, which tests/test_gpu_parity.py checks (Dense(H,1) is
// summed in a different order, so the outputs agree to ~1e-7).
//
// Also in this file, on the same building blocks:
//   * one-layer models (first = last: no ring, nobody waits) with FOUR HELPER WAVES that carry the whole DSP chain and
//     the per-stream scalar work of a frame — the whole run() in the one launch (lp_helper);
//   * stacked models in one launch: the packed chain passes on two waves of the first / last layer's workgroup around
//     the unchanged body (lp_chain_rows);
//   * k_gru_gm: one-layer GRUs on gate-major tiles (three quarters of the MFMAs), one main + one helper wave per SIMD.
#include <cstdlib>
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLpChunk = 256;            // frames of audio staged in LDS at a time
#ifndef AIDAX_LP_RING
#define AIDAX_LP_RING 32                 // (measurement builds: -DAIDAX_LP_RING=16 -DAIDAX_LP_BATCH=4, scratch/mkvariant_lp.sh)
#endif
#ifndef AIDAX_LP_BATCH
#define AIDAX_LP_BATCH 8
#endif
constexpr int kLpRing = AIDAX_LP_RING;   // frames of h in flight between two layers
constexpr int kLpBatch = AIDAX_LP_BATCH; // frames per counter update (a release costs ~2-6 us, an acquire ~1.7 us: MI355X_MICROARCH.md)
// A waiting workgroup gives up after kLpGiveUpTicks of the 100 MHz wall clock (250 ms: far beyond any wait that co-tenant
// kernels can cause while the partner still gets a CU — a launch must end even if its partner never runs). The pass is then
// wrong; the workgroup says so in the pool's fault word (pinned host memory), the pool reports AIDAX_ERR_DEVICE for it
// and serves the model with k_mfma from then on (aidax_pool.cpp).
constexpr uint64_t kLpGiveUpTicks = 25000000ull;
__device__ __forceinline__ bool lp_timed_out(uint32_t& spins, uint64_t& t0)
{
    if (spins++ == 0) { t0 = wall_clock64(); return false; }
    return (spins & 63u) == 0 && wall_clock64() - t0 > kLpGiveUpTicks;
}

__host__ __device__ inline size_t lp_lds_floats(int hidden, int n_frames, int n_helpers = 0)
{
    const size_t nP = (size_t)(((n_frames < kLpChunk ? n_frames : kLpChunk) + 3) & ~3);
    return (size_t)kMfmaStreams * nP                          /* xb: audio rows (first and last layer)   */
         + 2 * 64                                             /* xin[parity][4][n]  (first layer)        */
         + (size_t)2 * hidden * kMfmaStreams                  /* below[parity][unit][n] (layers >= 1)    */
         + (size_t)2 * hidden * kMfmaStreams                  /* hT[parity][unit][n]                     */
         + (size_t)hidden * kMfmaStreams                      /* cT[unit][n]                             */
         + (size_t)((hidden + 1 + 3) & ~3)                    /* Dense weights + bias (last layer)       */
         + kMfmaStreams                                       /* live flags                              */
         + 2 * 8 * kMfmaStreams                               /* Dense partial sums [parity][wave][n]    */
         + (size_t)n_helpers * 2 * kChainHandFloats           /* one-launch form: hand-over slots of the helper waves' pre and post pass */
#ifdef AIDAX_LP_TRACE
         + 2048                                               /* scratch/lp_trace.py: time stamps of eight ticks, 12 waves, 8 stamps (u64) needs 1536 floats */
#endif
         ;
}

// one frame in the ring: h [unit][stream], then (M > 0) the pre-activations of the first M tiles of every wave of the
// layer above, started by the layer below ([wave][tile][lane] x 4 gate rows)
__host__ __device__ constexpr size_t lp_slot_floats(int hidden, int waves, int m) { return (size_t)hidden * kMfmaStreams + (size_t)waves * m * kWave * 4; }
__host__ __device__ constexpr size_t lp_ring_floats(int hidden, int waves, int m) { return (size_t)kLpRing * lp_slot_floats(hidden, waves, m); }
// Tiles per wave whose input half moves from the upper layer's workgroup to the lower one's. Two layers: the upper
// has 8G k-steps per tile, the lower 4G + 1 — moving one tile's input half (4G MFMAs per wave) evens them out
// (LSTM-96: 75 / 144 -> 99 / 120 MFMAs per wave and frame). Deeper stacks are bound by their last layer either way.
// LSTM-96 (three tiles per wave): one moved tile still leaves 99 / 120; the waves of a workgroup's FIRST half move two
// and those of the second half one (waves w and w + NW/2 share a SIMD, so every SIMD carries one of each): 111 / 108.
// Returned as the LARGEST count a wave moves: 2 means "two for waves < NW/2, one for the others".
// (Only with eight waves: four waves sit on a SIMD each, and an uneven split would just make the slowest wave slower.)
__host__ __device__ constexpr int lp_moved_tiles(int n_layers, int tpw, int nw) { return n_layers != 2 ? 0 : (nw == 8 && tpw == 3) ? 2 : tpw >= 2 ? 1 : 0; }
constexpr int kLpCounterStride = 32;     // uint32 per (group, boundary): produced at [0], consumed at [16] (own cache lines)

// 16-byte device-scope (sc0 sc1) accesses: write-through stores / L1-bypassing loads. Buffer instructions, so the
// compiler keeps track of their completion (vmcnt) like of any other load.
typedef unsigned lp_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kLpDeviceScope = 17;                         // aux bits: sc0 | sc1
#ifndef AIDAX_LP_LOAD_AUX
#define AIDAX_LP_LOAD_AUX kLpDeviceScope                   // (measurement builds: the ring's loads / stores at another scope — 16 = sc1, 1 = sc0, 0 = none)
#endif
#ifndef AIDAX_LP_STORE_AUX
#define AIDAX_LP_STORE_AUX kLpDeviceScope
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lp_rsrc(const float* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 lp_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AIDAX_LP_LOAD_AUX));
}
__device__ __forceinline__ void lp_store16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lp_u32x4, v), r, byte_off, 0, AIDAX_LP_STORE_AUX);
}

__device__ __forceinline__ float lp_row_sum16(float v)
{
    v = v + dpp_take<0xB1, 0xf>(v);
    v = v + dpp_take<0x4E, 0xf>(v);
    v = v + dpp_take<0x141, 0xf>(v);
    v = v + dpp_take<0x140, 0xf>(v);
    return v;
}

// acc[tl] += A(groups G0..G0+NG) . B, A resident in `wres`, B fragments from `src` ([unit][stream] in LDS).
// The B values of group g+1 are read while the MFMAs of group g issue (two register sets alternate by the parity of
// g under the full unroll): left to itself the compiler reads a pair of values, waits for the LDS, issues six MFMAs,
// and exposes the LDS latency 24 times per layer half.
template <int TPW, int NG, int G0, int NRES, int TL0 = 0, int TL1 = TPW, int NACC = TPW>
__device__ __forceinline__ void lp_gates(f32x4 (&acc)[NACC], const f32x4 (&wres)[NRES], const float* src, int lane)
{
    if constexpr (TL0 < TL1) {
        float b0[4], b1[4];
        auto read_b = [&](float (&b)[4], int g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = src[256 * g + 64 * j + lane];
        };
        auto group = [&](const float (&b)[4], int g) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tl = TL0; tl < TL1; ++tl) {
                    const int e = j * TPW + tl;
                    acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[(G0 + g) * TPW + e / 4][e % 4], b[j], acc[tl], 0, 0, 0);
                }
        };
        read_b(b0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g & 1) {
                if (g + 1 < NG) read_b(b0, g + 1);
                __builtin_amdgcn_sched_barrier(0);
                group(b1, g);
            } else {
                if (g + 1 < NG) read_b(b1, g + 1);
                __builtin_amdgcn_sched_barrier(0);
                group(b0, g);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifdef AIDAX_LP_TRACE
#define LP_STAMP(k) do { if ((int)blockIdx.x == ((AIDAX_TUNE(a) >> 16) & 0xff) && tick >= kTraceT0 && tick < kTraceT0 + 8) {                         \
        __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); __builtin_amdgcn_sched_barrier(0); \
        if ((threadIdx.x & 63) == 0) trace[((tick - kTraceT0) * 12 + (threadIdx.x >> 6)) * 8 + (k)] = t_; } } while (0)
#else
#define LP_STAMP(k) do {} while (0)
#endif

// ---------------------------------------------------------------- one-launch form: the helper waves
// One-layer models whose main waves leave room for a third wave per SIMD (lp_helpers) run their whole run() in this
// launch: NHELP extra waves, each the keeper of 16 / NHELP streams of the group in 8-lane groups like k_chain's. A helper
//   * runs the PRE pass (LPF -> pre-gain ramp -> EQ-pre) on its streams' rows in LDS ahead of the frame loop — as many
//     macro-steps of eight frames as the cascade is deep before the first tick, one more per tick until the chunk is done;
//   * writes the model inputs of the next frame (x * in_gain, the PARAM smoothers' samples) and finishes the Dense of
//     the frame before last (partial sums, bias, skip, output gain) — what lane < 16 of wave 0 did at the head of every
//     tick while seven waves waited (scratch/lp_trace.py: 2 100 cycles against 460);
//   * runs the POST pass (DC blocker -> EQ-post -> master ramp) over the finished frames, one macro-step whenever eight
//     more are there, the rest after the last tick, and stores the rows.
// The passes are k_chain's own (chain_macro_step): same operations per sample and stage, bit-identical state. The helper
// meets the main waves at every barrier of lp_body — their barrier sequences must stay the same.
template <int H, int NW, int NHELP>
__device__ __forceinline__ void lp_helper(const LaunchArgs& a, float* xb, float* xin, const float* wdl, float* livef,
                                          const float* dpart, float* hands, int grp, int nP)
{
    constexpr int NS = kMfmaStreams;
    constexpr int SPH = NS / NHELP;                        // streams per helper wave
    static_assert(SPH * NHELP == NS && SPH <= 8, "eight lanes per stream");
    if (!(AIDAX_TUNE(a) & 1024)) __builtin_amdgcn_s_setprio(3);   // (measurement: AIDAX_TUNE bit 1024 leaves the helpers at the main waves' priority)
    const int lane = threadIdx.x & 63;
    const int hw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - NW;
    const int gl = lane >> 3, stage = lane & 7;
    const int n = (int)a.n_frames;
    const int I = a.input_size;
    const int s_base = grp * NS;
    const bool slot_ok = gl < SPH;                         // this lane group stands for stream slot `sl` of the group
    const int sl = hw * SPH + (slot_ok ? gl : 0);
    const bool valid = slot_ok && s_base + sl < (int)a.n_streams;
    const int sg = valid ? s_base + sl : s_base;           // the others shadow the group's first stream and never store
    const bool keeper = slot_ok && stage == 0;             // the lane that does the per-stream scalar work

    const StreamCtl& ctl = a.ctl[sg];
    StreamState& st = a.st[sg];
    const uint32_t flags = ctl.flags;
    const uint32_t pending0 = st.pending;
    const bool live = valid && n != 0 && (flags & CTL_ENABLED);                 // :607-619
    const bool net = live && (flags & CTL_NET_ON);                             // :631-632

    float p_mem[2] = { st.p_mem[0], st.p_mem[1] }, p_tgt[2] = { st.p_tgt[0], st.p_tgt[1] }, p_step[2] = { st.p_step[0], st.p_step[1] };
    uint32_t pending = pending0 & ~PEND_ACTIVATE;
    if (net) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                      // LinearValueSmoother::setTargetValue (:209-216)
            const float nt = ctl.p_target[i];
            if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
                p_tgt[i] = nt;
                p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
            }
        }
        if (pending & PEND_PARAM_FIRST) {                  // paramFirstRun (:636-640)
            pending &= ~PEND_PARAM_FIRST;
            p_mem[0] = p_tgt[0];
            p_mem[1] = p_tgt[1];
        }
    }
    if (keeper) livef[sl] = net ? 1.f : 0.f;

    // the two passes of this lane's stage, set up like k_chain<true> / k_chain<false>
    ChainPass P, Q;
    float pre_mem0, pre_tgt, master_mem0, master_tgt;
    int slot_p, slot_q;
    {
        const bool eq = flags & CTL_EQ_PRE;
        P.K = eq ? 6 : 1;
        P.gain_lane = 0;
        const int k = stage < P.K ? stage : 0;
        slot_p = pre_slot(k);
        const bool act = k == 0 ? (flags & CTL_LPF_ON) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot_p == BQ_MID : true);
        chain_load(P, ctl, st, slot_p, act);
        pre_mem0 = st.pre_mem; pre_tgt = st.pre_tgt;
        if (pending0 & PEND_ACTIVATE) pre_mem0 = pre_tgt;  // activate(): clearToTargetValue (:341-342)
        pre_tgt = ctl.pre_target;                          // :513, before every early-out
        P.g.arm(pre_mem0, pre_tgt, ctl.pre_coef);
    }
    {
        const bool eq = flags & CTL_EQ_POST;
        Q.K = eq ? 6 : 1;
        Q.gain_lane = Q.K - 1;
        const int k = stage < Q.K ? stage : 0;
        slot_q = post_slot(k);
        const bool act = k == 0 ? (flags & CTL_DC_ON) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot_q == BQ_MID : true);
        chain_load(Q, ctl, st, slot_q, act);
        master_mem0 = st.master_mem; master_tgt = st.master_tgt;
        if (pending0 & PEND_ACTIVATE) master_mem0 = master_tgt;
        if (live) master_tgt = ctl.master_target;          // :654, only on the DSP path
        Q.g.arm(master_mem0, master_tgt, ctl.master_coef);
    }
    const bool run_p = live && stage < P.K, run_q = live && stage < Q.K;
    const bool general = (AIDAX_TUNE(a) & 8) != 0;
    const int depth_p = general || __builtin_amdgcn_ballot_w64(run_p && stage > 0) != 0 ? 6 : 1;
    const int depth_q = general || __builtin_amdgcn_ballot_w64(run_q && stage > 0) != 0 ? 6 : 1;
    float* row = xb + sl * nP;
    float* hand_p = hands + (size_t)(2 * hw) * kChainHandFloats;
    float* hand_q = hand_p + kChainHandFloats;
#ifdef AIDAX_LP_TRACE
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(hands + NHELP * 2 * kChainHandFloats);
    constexpr int kTraceT0 = 96;
#endif
    __syncthreads();                                       // (1) set-up

    auto write_xin = [&](int parity, int f) {              // x * in_gain, PARAM1, PARAM2 of frame f (:171-181, :195-231)
        float q1 = 0.f, q2 = 0.f;
        if (I >= 2) q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
        if (I >= 3) q2 = lin_next(p_mem[1], p_tgt[1], p_step[1]);
        if (keeper) {
            float* col = xin + parity * 64;
            col[sl] = net ? row[f] * a.in_gain : 0.f;
            col[NS + sl] = q1;
            col[2 * NS + sl] = q2;
            col[3 * NS + sl] = 0.f;
        }
    };

    for (int base = 0; base < n; base += kLpChunk) {
        const int cnt = n - base < kLpChunk ? n - base : kLpChunk;
        const int n_full = cnt & ~(kChainBlock - 1);
        __syncthreads();                                   // (2) the chunk's rows are in LDS
        ChainJob jp, jq;
        chain_job_begin(jp, P, stage, run_p, n_full, general);
        bool p_open = true;
        auto pre_work = [&](int steps) {                   // the ragged tail right behind the last whole block
            for (int k = 0; k < steps && chain_job_pending(jp, depth_p); ++k) chain_job_step(jp, P, stage, run_p, row, hand_p, lane);
            if (p_open && !chain_job_pending(jp, depth_p)) {
                chain_job_end(jp, P, run_p);
                if (n_full != cnt) chain_sweep<1>(P, stage, run_p, depth_p, row + n_full, row + n_full, cnt - n_full);
                p_open = false;
            }
        };
        pre_work(depth_p);                                 // frames 0..7 (and what a short chunk has) are final: block 0 has left the cascade after `depth` macro-steps
        chain_job_begin(jq, Q, stage, run_q, n_full, general);
        write_xin(0, 0);
        __syncthreads();                                   // (3)
        const int ticks = cnt + 2;
        for (int tick = 0; tick < ticks; ++tick) {
            LP_STAMP(0);
            if (tick >= 2 && keeper) {                     // Dense of frame tick-2: the waves' partial sums, bias, skip, gain
                const int fd = tick - 2;
                float y = wdl[H];
#pragma unroll
                for (int w = 0; w < NW; ++w) y += dpart[(((tick - 1) & 1) * NW + w) * NS + sl];
                const float x = row[fd] * a.in_gain;
                float o = a.input_skip ? x + y : y;
                o = o * a.out_gain;
                if (net) row[fd] = o;
            }
            __builtin_amdgcn_wave_barrier();
            // frames 0 .. tick-2 of the rows are model output now: a post-pass step when eight more are there
            if (chain_job_pending(jq, depth_q) && chain_job_needs(jq) <= tick - 1) chain_job_step(jq, Q, stage, run_q, row, hand_q, lane);
            pre_work(1);
            if (tick + 1 < cnt) write_xin((tick + 1) & 1, tick + 1);
            LP_STAMP(4);
            __syncthreads();                               // the tick's barrier
            LP_STAMP(5);
        }
        while (chain_job_pending(jq, depth_q)) chain_job_step(jq, Q, stage, run_q, row, hand_q, lane);
        chain_job_end(jq, Q, run_q);
        if (n_full != cnt) chain_sweep<1>(Q, stage, run_q, depth_q, row + n_full, row + n_full, cnt - n_full);
        __builtin_amdgcn_wave_barrier();
        // every valid row goes out: processed, or — a disabled stream — the raw input (the hard bypass copy of :612-619)
        for (int g = 0; g < SPH; ++g) {
            const int s2 = s_base + hw * SPH + g;
            if (s2 >= (int)a.n_streams) continue;
            if ((a.ctl[s2].flags & CTL_ENABLED) || a.out != a.in) {
                float* dst = a.out + (size_t)s2 * n + base;
                const float* src = xb + (hw * SPH + g) * nP;
                if (((n | base) & 3) == 0) store_block(dst, src, cnt, lane);
                else for (int t = lane; t < cnt; t += kWave) dst[t] = src[t];
            }
        }
        __syncthreads();                                   // (4) the rows may be overwritten
    }
#ifdef AIDAX_LP_TRACE
    if ((int)blockIdx.x == ((AIDAX_TUNE(a) >> 16) & 0xff)) __syncthreads();
#endif
    __syncthreads();                                       // (5)
    if (!valid) return;
    if (run_p) { st.z[slot_p][0] = P.z1; st.z[slot_p][1] = P.z2; }
    if (run_q) { st.z[slot_q][0] = Q.z1; st.z[slot_q][1] = Q.z2; }
    if (stage == 0) { st.pre_mem = live ? P.g.mem : pre_mem0; st.pre_tgt = pre_tgt; }
    if (stage == Q.gain_lane) { st.master_mem = live ? Q.g.mem : master_mem0; st.master_tgt = master_tgt; }
    if (keeper) {
        st.pending = pending;
        if (net) {
            st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
            st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
            st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
        }
    }
}

// M: tiles per wave of the layer above whose input half the first layer computes (lp_moved_tiles; two-layer models).
// FIRST / LAST: the role of this workgroup's layer, a compile-time constant of the body — the kernel branches once on the
// layer index, so each role's registers are allocated for that role only (the first layer carries no fetched tiles, no
// Dense fragments; the others no started-tile accumulators): LSTM-96 x2 fits its 256 registers without scratch.
template <int TPW, int NW, int M, bool FIRST, bool LAST, int NHELP = 0, bool CHAIN = false>
__device__ __forceinline__ void lp_body(const LaunchArgs& a, const MfmaDesc& d, float* ring, uint32_t* counters, uint32_t* fault,
                                        float* smem, int grp, int l)
{
    constexpr int H = 4 * TPW * NW;
    constexpr int NT = NW * kWave;
    constexpr int NS = kMfmaStreams;
    constexpr int G = H / 16;                              // k-step groups per H columns
    constexpr int NRES = 2 * G * TPW;                      // f32x4 of resident A fragments: [h below | own h]
    constexpr int MA = M > 0 ? M : 1;                      // array extent of the moved-tile registers
    constexpr size_t kSlot = lp_slot_floats(H, NW, M);     // floats of one ring frame
    static_assert(NHELP == 0 || (FIRST && LAST), "helper waves serve one-layer models");
    constexpr bool fused = NHELP > 0;                      // the whole run() in this launch: a.in -> a.out, MODE_CHAIN only
    // CHAIN: the kernel runs the DSP chain around this body (lp_chain_rows); the body itself is the same as between two
    // k_chain launches, except that it clears its bit of StreamState::pending with an atomic.
    constexpr bool chain = CHAIN;
    static_assert(!(CHAIN && NHELP > 0), "one form or the other");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = M == 2 ? (wave < NW / 2 ? 2 : 1) : M;   // tiles THIS wave moves (M == 2: the first half of the waves two, the others one)
    const int n = (int)a.n_frames;
    const int NL = d.n_layers;
    const int Ht = d.hidden_true;
    const int I = a.input_size;
    const int mode = a.mode;
    const int blk = (int)blockIdx.x;
    constexpr bool first = FIRST, last = LAST;
    const int s_base = grp * NS;
    const int chunk = n < kLpChunk ? n : kLpChunk;
    const int nP = (chunk + 3) & ~3;

    float* xb    = smem;                                   // [NS][nP]
    float* xin   = xb + NS * nP;                           // [2][4][NS]
    float* below = xin + 2 * 64;                           // [2][H][NS]
    float* hT    = below + 2 * H * NS;                     // [2][H][NS]
    float* cT    = hT + 2 * H * NS;                        // [H][NS]
    float* wdl   = cT + H * NS;                            // Dense weights, bias at [H]
    float* livef = wdl + ((H + 1 + 3) & ~3);               // [NS]
    float* dpart = livef + NS;                             // [2][NW][NS] Dense partial sums of the waves (last layer)
#ifdef AIDAX_LP_TRACE
    // measurement build only (scratch/lp_trace.py): workgroup 0 stamps the shader clock at six points of ticks 96..103
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(dpart + 2 * 8 * NS + NHELP * 2 * kChainHandFloats);
    constexpr int kTraceT0 = 96;
#endif

    if constexpr (fused) {
        if (wave >= NW) {
            lp_helper<H, NW, NHELP>(a, xb, xin, wdl, livef, dpart, dpart + 2 * 8 * NS, grp, nP);
            return;
        }
    }
    const float* W = a.wpack;
    const MfmaLayer& L = d.L[l];

    // ---- per-stream bookkeeping: lanes tid < NS own stream s_base+tid (live flag; PARAM smoothers on the first layer)
    float p_mem[2] = { 0.f, 0.f }, p_tgt[2] = { 0.f, 0.f }, p_step[2] = { 0.f, 0.f };
    uint32_t pending = 0, st_pending0 = 0;
    bool mine_live = false;
    if (!fused && tid < NS) {
        const int sg = s_base + tid;
        const bool valid = sg < (int)a.n_streams;
        if (valid) {
            StreamState& st = a.st[sg];
            p_mem[0] = st.p_mem[0]; p_mem[1] = st.p_mem[1];
            p_tgt[0] = st.p_tgt[0]; p_tgt[1] = st.p_tgt[1];
            p_step[0] = st.p_step[0]; p_step[1] = st.p_step[1];
            pending = st_pending0 = st.pending;
            if (mode == MODE_CHAIN) {
                const StreamCtl& ctl = a.ctl[sg];
                const uint32_t flags = ctl.flags;
                mine_live = n != 0 && (flags & CTL_ENABLED) && (flags & CTL_NET_ON);      // :607-619, :631-632
                if (mine_live) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {            // LinearValueSmoother::setTargetValue (:209-216)
                        const float nt = ctl.p_target[i];
                        if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
                            p_tgt[i] = nt;
                            p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
                        }
                    }
                    if (pending & PEND_PARAM_FIRST) {        // paramFirstRun (:636-640)
                        pending &= ~PEND_PARAM_FIRST;
                        p_mem[0] = p_tgt[0];
                        p_mem[1] = p_tgt[1];
                    }
                }
            } else {
                mine_live = n != 0 && (mode == MODE_WARMUP || sg == 0);
            }
        }
        livef[tid] = mine_live ? 1.f : 0.f;
    }
    if (last) for (int i = tid; i < H + 1; i += NT) wdl[i] = W[d.wd_off + i];
    // this layer's recurrent state -> LDS (parity 0 is what frame 0 reads)
    for (int i = tid; i < H * NS; i += NT) {
        const int u = i / NS, sn = i % NS, sg = s_base + sn;
        const bool valid = sg < (int)a.n_streams;
        const float* stp = a.nn + (size_t)(valid ? sg : 0) * a.nn_stride + L.state_off;
        hT[i] = (valid && u < Ht) ? stp[u] : 0.f;           // padded units rest at 0
        cT[i] = (valid && u < Ht && L.cell == 0) ? stp[Ht + u] : 0.f;
    }

    // ---- this layer's bias rows (the accumulators start from them) and A fragments, resident for the launch
    f32x4 bias_r[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
        bias_r[tl] = *reinterpret_cast<const f32x4*>(W + L.b_off + 4 * (4 * (wave * TPW + tl) + (lane >> 4)));
    // One register set, two uses that exclude each other: on the first layer the bias rows the moved tiles start from
    // (the upper layer's), on the others the started tiles of the NEXT frame on their way in from the ring.
    f32x4 upx[MA];
#pragma unroll
    for (int tl = 0; tl < MA; ++tl)
        upx[tl] = (M > 0 && first) ? *reinterpret_cast<const f32x4*>(W + d.L[M > 0 ? 1 : 0].b_off + 4 * (4 * (wave * TPW + tl) + (lane >> 4)))
                                    : f32x4{ 0.f, 0.f, 0.f, 0.f };
    // Dense(H,1) on the matrix cores (last layer): y = wd . h is one more 16-row tile whose row 0 is wd; its H/4
    // k-steps are dealt out TPW per wave, each wave leaves a partial sum per stream in LDS and wave 0 adds the NW
    // partials in a fixed order a tick later. (As VALU code — six LDS reads with bank conflicts, a DPP tree, twice per
    // SIMD — it cost ~1000 cycles per tick next to the MFMA stretch; this way it is TPW MFMAs per wave.)
    float dfrag[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
        dfrag[i] = (last && (lane & 15) == 0) ? W[d.wd_off + 4 * (wave * TPW + i) + (lane >> 4)] : 0.f;
    float w_in0[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) w_in0[tl] = first ? W[d.L[0].w_in_off + ((size_t)wave * kWave + lane) * TPW + tl] : 0.f;
    f32x4 wres[NRES];
    {
        const int g_tot = first ? G : 2 * G;
        const f32x4* ap = reinterpret_cast<const f32x4*>(W + L.w_big_off) + ((size_t)wave * g_tot * kWave + lane) * TPW;
        const f32x4* ap_up = reinterpret_cast<const f32x4*>(W + d.L[M > 0 ? 1 : 0].w_big_off) + ((size_t)wave * 2 * G * kWave + lane) * TPW;
#pragma unroll
        for (int g = 0; g < 2 * G; ++g)
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                // first layer: its G groups (own h) sit in the upper half, like the recurrent half of the others; with
                // M > 0 its lower half holds the layer ABOVE's input-half fragments (same wave, same lane): it starts
                // that layer's first M tiles
                const int gs = first ? g - G : g;
                if (gs >= 0) wres[g * TPW + q] = ap[(size_t)gs * kWave * TPW + q];
                else if (M > 0) wres[g * TPW + q] = ap_up[(size_t)g * kWave * TPW + q];
                else wres[g * TPW + q] = f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
    }

    // ---- ring bookkeeping. Counters count frames since the buffers were allocated and are equal on both sides
    // between launches, so a workgroup starts from its own side's value.
    const size_t ring_stride = lp_ring_floats(H, NW, M);
    float* ring_out = last ? nullptr : ring + ((size_t)grp * (NL - 1) + l) * ring_stride;
    const float* ring_in = first ? nullptr : ring + ((size_t)grp * (NL - 1) + (l - 1)) * ring_stride;
    const size_t ring_bytes = ring_stride * sizeof(float);
    const __amdgpu_buffer_rsrc_t rs_out = lp_rsrc(last ? ring : ring_out, ring_bytes);
    const __amdgpu_buffer_rsrc_t rs_in = lp_rsrc(first ? ring : ring_in, ring_bytes);
    uint32_t* cnt_out = last ? nullptr : counters + ((size_t)grp * (NL - 1) + l) * kLpCounterStride;
    uint32_t* cnt_in = first ? nullptr : counters + ((size_t)grp * (NL - 1) + (l - 1)) * kLpCounterStride;
    auto give_up = [&]() { __hip_atomic_fetch_add(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    if ((AIDAX_TUNE(a) & 16) && blk == 0 && tid == 0) give_up();      // test hook: report a give-up that did not happen
    // every thread needs the bases (they place a frame in the ring); the running counts are thread 0's business
    const uint32_t base_out = cnt_out ? __hip_atomic_load(cnt_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t base_in = cnt_in ? __hip_atomic_load(cnt_in + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t known_free = kLpRing;                         // frames of this launch the ring above is known to have room for
    uint32_t known_below = 0;                              // frames of this launch known to exist below
    __syncthreads();

    // Frame Fprev goes up the ring, one tick after it was computed (h_src = h(Fprev), stable in LDS for this tick):
    // h as 16-byte write-through stores, and (first layer, M > 0) the started tiles of the layer above:
    // W_in(above)[first M tiles] . h(Fprev) + bias.
    constexpr int kHVec = H * NS / 4;                       // f32x4 of one h tile
    auto ship_frame = [&](const float* h_src, int Fprev) {
        const uint32_t slot_off = (uint32_t)(((base_out + (uint32_t)Fprev) % kLpRing) * kSlot * sizeof(float));
        for (int i = tid; i < kHVec; i += NT)
            lp_store16(rs_out, slot_off + (uint32_t)i * 16u, reinterpret_cast<const f32x4*>(h_src)[i]);
        if constexpr (M > 0) {
            if (first) {
                f32x4 pacc[MA];
#pragma unroll
                for (int tl = 0; tl < M; ++tl) pacc[tl] = upx[tl];
                if constexpr (M == 2) {
                    if (mw == 2) lp_gates<TPW, G, 0, NRES, 0, 2, MA>(pacc, wres, h_src, lane);
                    else lp_gates<TPW, G, 0, NRES, 0, 1, MA>(pacc, wres, h_src, lane);
                } else {
                    lp_gates<TPW, G, 0, NRES, 0, M, MA>(pacc, wres, h_src, lane);
                }
#pragma unroll
                for (int tl = 0; tl < M; ++tl)
                    if (tl < mw)
                        lp_store16(rs_out, slot_off + (uint32_t)(H * NS * sizeof(float)) + (uint32_t)(((wave * M + tl) * kWave + lane) * 16), pacc[tl]);
            }
        }
    };
    // the cell state of this lane's (unit, stream) pairs — one per tile — lives in registers for the launch (cT is only how
    // it travels between the state records and the lanes)
    float creg[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) creg[tl] = cT[(wave * TPW + tl) * 64 + lane];
    // frames [0, total) of this launch as the ring sees them
    int par = 0;                                           // parity of hT the next frame reads
    int done = 0;                                          // frames finished before this chunk
    for (int base = 0; base < n; base += kLpChunk) {
        const int cnt = n - base < kLpChunk ? n - base : kLpChunk;
        // what the layer below hands over: frame F of the launch sits in ring slot (base_in + F) % kLpRing
        auto wait_below = [&](int frames_needed) {        // thread 0 only: until `frames_needed` frames of the launch exist
            if (frames_needed > n) frames_needed = n;
            uint32_t spins = 0;
            uint64_t t0 = 0;
            while ((int)known_below < frames_needed) {
                known_below = __hip_atomic_load(cnt_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base_in;
                if ((int)known_below < frames_needed) {
                    if (lp_timed_out(spins, t0)) { give_up(); known_below = (uint32_t)n; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        };
        // ---- audio rows of this chunk: the first layer reads them as input, the last for in_skip and to deliver
        if constexpr (chain && last && !first) {
            // one-launch form: the rows in a.out are the pre pass's, stored (write-through, drained) by the FIRST layer's
            // workgroup before it computed its first frame — once frames exist below, the rows are there
            if (tid == 0) wait_below(done + 2);
            __syncthreads();
            // ... and must come from L2: the rows of neighbouring stream groups can share a cache line (odd block lengths),
            // and another workgroup on this CU may have brought that line in before the pre pass stored into it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (first || last) {
            for (int sl = wave; sl < NS; sl += NW) {
                const int sg = s_base + sl;
                const bool lv = fused ? sg < (int)a.n_streams : livef[sl] != 0.f;     // one-launch form: every row, the chains run on net-off streams too
                float* row = xb + sl * nP;
                if (fused) {
                    const float* src = a.in + (size_t)(lv ? sg : 0) * n + base;
                    if (lv && ((n | base) & 3) == 0) load_block(row, src, cnt, lane);
                    else for (int t = lane; t < cnt; t += kWave) row[t] = lv ? src[t] : 0.f;
                } else if (lv && mode == MODE_CHAIN) {
                    const float* src = a.out + (size_t)sg * n + base;
                    if (((n | base) & 3) == 0) load_block(row, src, cnt, lane);
                    else for (int t = lane; t < cnt; t += kWave) row[t] = src[t];
                } else if (lv && mode == MODE_NN_ONLY) {
                    for (int t = lane; t < cnt; t += kWave) row[t] = a.in[(size_t)(base + t) * I];
                } else {
                    for (int t = lane; t < cnt; t += kWave) row[t] = 0.f;
                }
            }
        }
        __syncthreads();
        auto write_xin = [&](int parity, int f) {
            // lanes tid < NS: x * in_gain, PARAM1, PARAM2 of frame `f` (:171-181, :195-231)
            float q1 = 0.f, q2 = 0.f;
            if (mode == MODE_CHAIN) {
                if (I >= 2) q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
                if (I >= 3) q2 = lin_next(p_mem[1], p_tgt[1], p_step[1]);
            } else if (mode == MODE_WARMUP) {             // constant params over the zero pre-buffer (:1077-1078)
                q1 = I >= 2 ? p_mem[0] : 0.f;
                q2 = I >= 3 ? p_mem[1] : 0.f;
            } else if (tid == 0) {
                q1 = I >= 2 ? a.in[(size_t)(base + f) * I + 1] : 0.f;
                q2 = I >= 3 ? a.in[(size_t)(base + f) * I + 2] : 0.f;
            }
            float* col = xin + parity * 64;
            col[tid] = xb[tid * nP + f] * a.in_gain;
            col[NS + tid] = q1;
            col[2 * NS + tid] = q2;
            col[3 * NS + tid] = 0.f;
        };
        constexpr int PER4 = (kHVec + NT - 1) / NT;        // f32x4 of a frame each thread moves
        f32x4 pre[PER4];
        f32x4 pcur[MA] = {};                               // started tiles of the current frame (layers >= 1, M > 0); the next frame's: upx
        auto fetch_below = [&](int F) {                    // all threads: frame F of the launch -> registers
            const uint32_t slot_off = (uint32_t)(((base_in + (uint32_t)F) % kLpRing) * kSlot * sizeof(float));
#pragma unroll
            for (int q = 0; q < PER4; ++q)
                if (q * NT + tid < kHVec) pre[q] = lp_load16(rs_in, slot_off + (uint32_t)(q * NT + tid) * 16u);
            if constexpr (M > 0) {                         // ... and this wave's started tiles straight into registers
#pragma unroll
                for (int tl = 0; tl < M; ++tl)
                    if (tl < mw)
                        upx[tl] = lp_load16(rs_in, slot_off + (uint32_t)(H * NS * sizeof(float)) + (uint32_t)(((wave * M + tl) * kWave + lane) * 16));
            }
        };
        auto stash_below = [&](int parity) {
#pragma unroll
            for (int q = 0; q < PER4; ++q)
                if (q * NT + tid < kHVec) reinterpret_cast<f32x4*>(below + parity * H * NS)[q * NT + tid] = pre[q];
        };

        if (first) {
            if (!fused && tid < NS) write_xin(0, 0);
        } else {
            if (tid == 0) wait_below(done + 2);            // frame 0 of the chunk for now, frame 1 for tick 0's prefetch
            __syncthreads();
            fetch_below(done);
            stash_below(0);
#pragma unroll
            for (int tl = 0; tl < MA; ++tl) pcur[tl] = upx[tl];
        }
        __syncthreads();

        const int ticks = last ? cnt + 2 : cnt;            // the Dense of a frame: partial sums one tick behind its h, the output two
        for (int tick = 0; tick < ticks; ++tick) {
            const int rd = par, wr = par ^ 1;
            const bool body = tick < cnt;
            const bool more = tick + 1 < cnt;              // another frame of this chunk follows
            const int F = done + tick;                     // frame of the launch this tick computes
            LP_STAMP(0);
            // ---- the next frame's input on its way: model inputs (first layer) / h of the layer below into
            // registers (others; thread 0 made sure of its existence before the previous barrier)
            if (first) {
                if (!fused && tid < NS && more) write_xin((tick + 1) & 1, tick + 1);
            } else if (more) {
                fetch_below(F + 1);
            }
            if (body && tid == 0) {
                // thread 0 looks ahead while the others compute: the frame AFTER next must exist below before the
                // next tick fetches it, and the slot of the next frame must be free above before the next tick
                // stores into it. Normally both are known already; otherwise this wave spins and the workgroup
                // waits for it at the barrier. (A counter read costs this wave ~1.5 us once per kLpBatch ticks — at the START
                // of the tick, where its SIMD partner has the matrix pipe to itself meanwhile. Reading the counters early
                // and looking at them just before the barrier was measured: 1 265 -> 1 305 us, nothing overlaps there.)
                if (!first && tick + 2 < cnt) wait_below(F + 3);
                if (!last && base + tick + 1 < n) {
                    uint32_t spins = 0;
                    uint64_t t0 = 0;
                    while ((int)known_free < F + 2) {
                        const uint32_t consumed = __hip_atomic_load(cnt_out + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base_out;
                        known_free = consumed + kLpRing;
                        if ((int)known_free < F + 2) {
                            if (lp_timed_out(spins, t0)) { give_up(); known_free = (uint32_t)n + kLpRing; break; }
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                }
            }

            // ---- Dense(H,1) of frame tick-1: this wave's k-steps against h(tick-1) (= hT[rd], stable for the whole tick).
            // At the head of the tick on purpose: its LDS reads and two MFMAs run while the gates' first operands are still on
            // their way from LDS. (Issued behind the gates, with the partial sum leaving after the cell update: cfg5 1 130 ->
            // 1 156 us, GRU-64 446 -> 466 us.)
            if (last && tick >= 1 && tick <= cnt) {
                f32x4 dacc = f32x4{ 0.f, 0.f, 0.f, 0.f };
                const float* hv = hT + rd * H * NS;
#pragma unroll
                for (int i = 0; i < TPW; ++i)
                    dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(dfrag[i], hv[64 * (wave * TPW + i) + lane], dacc, 0, 0, 0);
                if (lane < NS) dpart[((tick & 1) * NW + wave) * NS + lane] = dacc.x;      // row 0, column = stream `lane`
            }
            // ---- ... and of frame tick-2: the NW partial sums, bias, skip, output gain (wave 0, one lane per stream)
            if (!fused && last && tick >= 2 && wave == 0 && lane < NS) {
                const int fd = tick - 2;
                float y = wdl[H];
#pragma unroll
                for (int w = 0; w < NW; ++w) y += dpart[(((tick - 1) & 1) * NW + w) * NS + lane];
                const float x = xb[lane * nP + fd] * a.in_gain;
                float o = a.input_skip ? x + y : y;
                o = o * a.out_gain;
                if (livef[lane] != 0.f) xb[lane * nP + fd] = o;
            }

            LP_STAMP(1);
            if (body) {
                float* h_rd = hT + rd * H * NS;
                float* h_wr = hT + wr * H * NS;
                f32x4 acc[TPW];
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) acc[tl] = bias_r[tl];
                if (!last && F >= 1) ship_frame(h_rd, F - 1);          // h_rd = h(F-1)
                if (first) {                               // the model inputs: one k-step (x, PARAM1, PARAM2, 0)
                    const float b = xin[(tick & 1) * 64 + lane];
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl)
                        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in0[tl], b, acc[tl], 0, 0, 0);
                } else {
                    if constexpr (M > 0) {                 // the first M tiles arrive started (bias + input half), the others start here
#pragma unroll
                        for (int tl = 0; tl < M; ++tl)
                            if (tl < mw) acc[tl] = pcur[tl];
                    }
                    if constexpr (M == 2) {
                        if (mw == 2) lp_gates<TPW, G, 0, NRES, 2, TPW>(acc, wres, below + (tick & 1) * H * NS, lane);
                        else lp_gates<TPW, G, 0, NRES, 1, TPW>(acc, wres, below + (tick & 1) * H * NS, lane);
                    } else {
                        lp_gates<TPW, G, 0, NRES, M, TPW>(acc, wres, below + (tick & 1) * H * NS, lane);
                    }
                }
                LP_STAMP(2);
                lp_gates<TPW, G, G, NRES>(acc, wres, h_rd, lane);
                LP_STAMP(3);

#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) {
                    const int e = (wave * TPW + tl) * 64 + lane;          // unit 4T + (lane>>4), stream lane&15
                    float hn;
                    if (L.cell == 0) {
                        const float gi = sigmoid_pre(acc[tl].x), gf = sigmoid_pre(acc[tl].y);
                        const float gg = tanh_rat(acc[tl].z), go = sigmoid_pre(acc[tl].w);
                        const float cn = __builtin_fmaf(gf, creg[tl], gi * gg);
                        creg[tl] = cn;
                        hn = go * tanh_rat(cn);
                    } else {
                        const float gz = sigmoid_pre(acc[tl].x), gr = sigmoid_pre(acc[tl].y);
                        const float nn = tanh_exp_pre(__builtin_fmaf(gr, acc[tl].z, acc[tl].w));      // (GRU: see GruCell::step)
                        hn = __builtin_fmaf(gz, h_rd[e] - nn, nn);
                    }
                    h_wr[e] = hn;
                }
                par = wr;
                LP_STAMP(4);
                if (!first && more) {
                    stash_below((tick + 1) & 1);
#pragma unroll
                    for (int tl = 0; tl < MA; ++tl) pcur[tl] = upx[tl];
                }
                if (!last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's ring stores have left before the barrier (G16)
            }
            __syncthreads();                               // h(t) in LDS, the next frame's input in LDS, ring stores of this frame done
            LP_STAMP(5);
            // ---- counters, by thread 0 after the barrier: every kLpBatch frames and at the end of the launch
            if (body && tid == 0) {
                if (!last) {
                    // frames complete in the ring: a frame goes up one tick after it was computed (the last one
                    // after the loop); its write-through stores were drained before the barrier above
                    const int produced = F;
                    if (produced > 0 && produced % kLpBatch == 0)
                        __hip_atomic_store(cnt_out, base_out + (uint32_t)produced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!first) {
                    const int consumed = more ? F + 2 : F + 1;             // frames read out of the ring so far
                    if (consumed % kLpBatch == 0 || consumed == n)
                        __hip_atomic_store(cnt_in + 16, base_in + (uint32_t)consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        done += cnt;
        // ---- results of this chunk back to HBM (last layer)
        if (last && !fused) {
            for (int sl = wave; sl < NS; sl += NW) {
                const int sg = s_base + sl;
                if (livef[sl] == 0.f || mode == MODE_WARMUP) continue;
                float* dst = mode == MODE_CHAIN ? a.out + (size_t)sg * n + base : a.out + base;
                const float* row = xb + sl * nP;
                if (((n | base) & 3) == 0 && mode == MODE_CHAIN) store_block(dst, row, cnt, lane);
                else for (int t = lane; t < cnt; t += kWave) dst[t] = row[t];
            }
        }
        __syncthreads();
    }

    if (!last && n > 0) {                                  // the launch's last frame, then the final count
        ship_frame(hT + par * H * NS, n - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(cnt_out, base_out + (uint32_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

#ifdef AIDAX_LP_TRACE
    if ((int)blockIdx.x == ((AIDAX_TUNE(a) >> 16) & 0xff)) { __syncthreads(); for (int i = tid; i < 2 * 768; i += NT) fault[16 + i] = reinterpret_cast<const uint32_t*>(trace)[i]; }
#endif
    // ---- recurrent state and smoother memories back to HBM for the streams that ran
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) cT[(wave * TPW + tl) * 64 + lane] = creg[tl];
    __syncthreads();
    for (int i = tid; i < H * NS; i += NT) {
        const int u = i / NS, sn = i % NS, sg = s_base + sn;
        if (sg < (int)a.n_streams && u < Ht && livef[sn] != 0.f) {
            float* stp = a.nn + (size_t)sg * a.nn_stride + L.state_off;
            stp[u] = hT[par * H * NS + i];
            if (L.cell == 0) stp[Ht + u] = cT[i];
        }
    }
    if (!fused && first && tid < NS && mine_live && mode == MODE_CHAIN) {
        StreamState& st = a.st[s_base + tid];
        st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
        st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
        st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
        if (!chain) st.pending = pending;
        else if (pending != st_pending0)                  // (the last layer's workgroup clears PEND_ACTIVATE in the same word)
            __hip_atomic_fetch_and(&st.pending, ~(uint32_t)PEND_PARAM_FIRST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Stacked models, the one-launch form: their main waves hold the whole register file, so there is no room for helper waves
// — the packed chain passes (k_chain's body, chain_wave_pass) run on waves 0 and 1 of the first / last layer's workgroup
// BEFORE and AFTER lp_body, which stays exactly the body that runs between two k_chain launches (reshaping it for rows kept
// in LDS cost its frame loop 6 %: 1 096 -> 1 150 us on LSTM-96 x2 before a single chain instruction ran). The pre pass
// reads a.in and leaves its rows in a.out, where the body expects them — the LAST layer's workgroup needs them too (the model
// input for in_skip, and what a net-off stream delivers) and WAITS for them: the first layer's workgroup stores the rows
// write-through (device scope) and lets the stores drain before it computes its first frame, so the last layer's workgroup
// loads its rows only once frames exist in the ring below it (lp_body: the wait it has to do anyway, moved in front of
// the load). (Round 3's first version let the
// last layer's workgroup run the pre pass itself, uncommitted: it read the biquad states the first layer's workgroup was
// about to overwrite — right while the two started together, wrong once in 150 runs of the test suite.) The post pass
// takes the rows the body stored. Blocks of one staging chunk.
template <bool PRE>
__device__ __forceinline__ void lp_chain_rows(const LaunchArgs& a, float* smem, int grp, bool commit)
{
    constexpr int NS = kMfmaStreams;
    const int NT = (int)blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int nP = (n + 3) & ~3;
    const int s_base = grp * NS;
    float* rows = smem;                                    // [NS][nP]
    float* hands = rows + NS * nP;                         // [2][kChainPackHandFloats]
    const float* src_base = PRE ? a.in : a.out;
    // Device-scope loads, past this CU's vector cache. POST: the rows were stored by this workgroup a moment ago, and the
    // body's own read of them may have left their lines there, stale. PRE: with a.in == a.out the body would otherwise
    // find the lines this read brought in instead of the rows stored below.
    for (int i = tid; i < NS * n; i += NT) {
        const int sl = i / n, t = i - sl * n, sg = s_base + sl;
        rows[sl * nP + t] = sg < (int)a.n_streams ? __hip_atomic_load(src_base + (size_t)sg * n + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    __syncthreads();
    if (wave < 2)
        chain_wave_pass<PRE>(a, s_base + kChainWaveStreams * wave, rows + kChainWaveStreams * wave * nP, nP,
                             hands + wave * kChainPackHandFloats, n, lane, commit, !PRE);
    __syncthreads();
    // PRE: every valid row goes to out (a disabled stream's row is still the raw input: the hard bypass copy of :612-619);
    // POST: only rows that were processed
    for (int i = tid; i < NS * n; i += NT) {
        const int sl = i / n, t = i - sl * n, sg = s_base + sl;
        if (sg >= (int)a.n_streams) continue;
        const bool row_live = (a.ctl[sg].flags & CTL_ENABLED) != 0;
        if (PRE ? (row_live || a.out != a.in) : row_live)           // (device scope: another workgroup reads the pre pass's rows)
            __hip_atomic_store(a.out + (size_t)sg * n + t, rows[sl * nP + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                       // (waits for the stores: the body's loads come after them)
}

template <int TPW, int NW, int M, int NHELP = 0, bool CHAIN = false>
__global__ __launch_bounds__((NW + NHELP) * kWave) void k_mfma_lp(LaunchArgs a, MfmaDesc d, float* ring, uint32_t* counters, uint32_t* fault)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (NHELP > 0) {                             // one layer, the whole run() of its streams: workgroup = stream group
        const int n_groups = ((int)a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
        if ((int)blockIdx.x < n_groups) lp_body<TPW, NW, 0, true, true, NHELP>(a, d, ring, counters, fault, smem, (int)blockIdx.x, 0);
        return;
    }
    // workgroup id -> (group, layer): ids of one group are 8 apart, lower layers first
    // (AIDAX_TUNE bit 2 lays a group's layers out on ADJACENT ids instead — different XCDs under b % 8 — so that the
    // tests can exercise the cross-XCD hand-over too.)
    const int NL = d.n_layers;
    const int blk = (int)blockIdx.x;
    const bool adjacent = (AIDAX_TUNE(a) & 2) != 0;
    const int grp = adjacent ? blk / NL : (blk / (8 * NL)) * 8 + (blk & 7);
    const int l = adjacent ? blk % NL : (blk / 8) % NL;
    const int n_groups = ((int)a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    if (grp >= n_groups) return;
    if constexpr (CHAIN) {                                 // stacked models, and one-layer ones without room for helper waves
        if (NL == 1) {
            lp_chain_rows<true>(a, smem, grp, true);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            lp_body<TPW, NW, M, true, true, 0, true>(a, d, ring, counters, fault, smem, grp, l);
            __syncthreads();
            lp_chain_rows<false>(a, smem, grp, true);
        } else if (l == 0) {
            lp_chain_rows<true>(a, smem, grp, true);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // (the body's loads of the rows: from L2, see lp_body)
            lp_body<TPW, NW, M, true, false, 0, true>(a, d, ring, counters, fault, smem, grp, l);
        } else if (l == NL - 1) {
            if (AIDAX_TUNE(a) & 8192) {                                          // test hook: this workgroup starts 100 us late
                const uint64_t t0 = wall_clock64();
                while (wall_clock64() - t0 < 10000) __builtin_amdgcn_s_sleep(64);
            }
            lp_body<TPW, NW, M, false, true, 0, true>(a, d, ring, counters, fault, smem, grp, l);
            __syncthreads();
            lp_chain_rows<false>(a, smem, grp, true);
        } else {
            lp_body<TPW, NW, M, false, false, 0, true>(a, d, ring, counters, fault, smem, grp, l);
        }
        return;
    }
    if (NL == 1) lp_body<TPW, NW, M, true, true>(a, d, ring, counters, fault, smem, grp, l);      // one layer: no ring, nobody waits
    else if (l == 0) lp_body<TPW, NW, M, true, false>(a, d, ring, counters, fault, smem, grp, l);
    else if (l == NL - 1) lp_body<TPW, NW, M, false, true>(a, d, ring, counters, fault, smem, grp, l);
    else lp_body<TPW, NW, M, false, false>(a, d, ring, counters, fault, smem, grp, l);
}

// ================================================================ k_gru_gm: one-layer GRU, gate-major tiles
// The tiles of k_mfma / k_mfma_lp hold four gate rows per unit; a GRU has three that contract over h (z, r and the
// candidate's recurrent half) and one that only sees the model inputs, so a quarter of the recurrent MFMAs multiply zeros.
// Here a wave owns 16 units as three recurrent tiles of their own (z, r, recurrent half: H/4 k-steps each) plus the
// input half as a fourth accumulator that takes the input k-step only: 3 H/4 + 3 MFMAs per wave and frame instead of
// 4 (H/4 + 1) — GRU-64: 51 against 68 per SIMD. A lane then holds the four gates of FOUR units (rows 4q .. 4q+3 of each
// tile) for its stream, their h(t-1) stays in registers, and one main wave per SIMD does it (H / 16 of them) next to
// one helper wave (lp_helper: the DSP chain, the model inputs, the Dense's tail — the whole run() in this launch).
// Same weights and the same accumulation order per row as pack_mfma's other record: the recurrent state is
// bit-identical to k_mfma's (which warms the model up) and k_mfma_lp's.
__host__ __device__ inline size_t gm_lds_floats(int hidden, int n_frames, int n_helpers)
{
    const size_t nP = (size_t)(((n_frames < kLpChunk ? n_frames : kLpChunk) + 3) & ~3);
    return (size_t)kMfmaStreams * nP + 2 * 64 + (size_t)2 * hidden * kMfmaStreams + (size_t)((hidden + 1 + 3) & ~3) + kMfmaStreams
         + 2 * 8 * kMfmaStreams + (size_t)n_helpers * 2 * kChainHandFloats
#ifdef AIDAX_LP_TRACE
         + 2048                                             /* the helpers' time stamps land here (not dumped for this kernel) */
#endif
         ;
}

template <int UT, int NHELP>
__global__ __launch_bounds__((UT + NHELP) * kWave) void k_gru_gm(LaunchArgs a, MfmaDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * UT, NW = UT, NS = kMfmaStreams, KS = H / 4, NT = NW * kWave;
    constexpr int DK = KS / NW;                             // Dense k-steps per wave (= 4)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int grp = (int)blockIdx.x;
    const int s_base = grp * NS;
    const int Ht = d.hidden_true;
    const int chunk = n < kLpChunk ? n : kLpChunk;
    const int nP = (chunk + 3) & ~3;
    float* xb    = smem;                                    // [NS][nP] audio rows
    float* xin   = xb + NS * nP;                            // [2][4][NS] model inputs of a frame
    float* hT    = xin + 2 * 64;                            // [2][H][NS]
    float* wdl   = hT + 2 * H * NS;                         // Dense weights, bias at [H]
    float* livef = wdl + ((H + 1 + 3) & ~3);                // [NS]
    float* dpart = livef + NS;                              // [2][8][NS] Dense partial sums of the waves
    float* hands = dpart + 2 * 8 * NS;
    if (wave >= NW) {
        lp_helper<H, NW, NHELP>(a, xb, xin, wdl, livef, dpart, hands, grp, nP);
        return;
    }
    const float* W = a.wpack;
    const MfmaLayer& L = d.L[0];
    const int q = lane >> 4, c = lane & 15;
    for (int i = tid; i < H + 1; i += NT) wdl[i] = W[d.wd_off + i];
    // this wave's record (pack_mfma: gate-major)
    const float* rec = W + d.gm_off + (size_t)wave * kWave * (3 + 3 * KS + 16);
    float w_in[3], w_rec[KS][3];
#pragma unroll
    for (int g = 0; g < 3; ++g) w_in[g] = rec[g * kWave + lane];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int g = 0; g < 3; ++g) w_rec[kk][g] = rec[(3 + 3 * kk + g) * kWave + lane];
    f32x4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(rec + (3 + 3 * KS) * kWave + (g * kWave + lane) * 4);
    float dfrag[DK];
#pragma unroll
    for (int i = 0; i < DK; ++i) dfrag[i] = c == 0 ? W[d.wd_off + 4 * (wave * DK + i) + q] : 0.f;
    // h(t-1) of this lane's four units (16 wave + 4q + e) of stream s_base + c: registers for the launch, LDS for the others
    const int sg = s_base + c;
    const bool valid = sg < (int)a.n_streams;
    const float* stp = a.nn + (size_t)(valid ? sg : 0) * a.nn_stride + L.state_off;
    float hreg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int u = 16 * wave + 4 * q + e;
        hreg[e] = (valid && u < Ht) ? stp[u] : 0.f;         // padded units rest at 0
        hT[u * NS + c] = hreg[e];
    }
    __syncthreads();                                        // (1)

    int par = 0;
    for (int base = 0; base < n; base += kLpChunk) {
        const int cnt = n - base < kLpChunk ? n - base : kLpChunk;
        for (int sl = wave; sl < NS; sl += NW) {            // every valid row: the chains run on net-off streams too
            const int s2 = s_base + sl;
            const bool lv = s2 < (int)a.n_streams;
            float* row = xb + sl * nP;
            const float* src = a.in + (size_t)(lv ? s2 : 0) * n + base;
            if (lv && ((n | base) & 3) == 0) load_block(row, src, cnt, lane);
            else for (int t = lane; t < cnt; t += kWave) row[t] = lv ? src[t] : 0.f;
        }
        __syncthreads();                                    // (2)
        __syncthreads();                                    // (3) the helpers have run the head of the pre pass and written frame 0's inputs
        const int ticks = cnt + 2;
        for (int tick = 0; tick < ticks; ++tick) {
            const float* h_rd = hT + par * H * NS;
            float* h_wr = hT + (par ^ 1) * H * NS;
            if (tick >= 1 && tick <= cnt) {                 // Dense of frame tick-1: this wave's k-steps against h(tick-1)
                f32x4 dacc = f32x4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int i = 0; i < DK; ++i)
                    dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(dfrag[i], h_rd[64 * (wave * DK + i) + lane], dacc, 0, 0, 0);
                if (lane < NS) dpart[((tick & 1) * NW + wave) * NS + lane] = dacc.x;
            }
            if (tick < cnt) {
                f32x4 az = bias[0], ar = bias[1], an = bias[2], ax = bias[3];
                const float bx = xin[(tick & 1) * 64 + lane];
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[0], bx, az, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[1], bx, ar, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[2], bx, ax, 0, 0, 0);
                // h k-steps, the B values a group of four ahead of the MFMAs that use them
                float b0[4], b1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b0[j] = h_rd[64 * j + lane];
#pragma unroll
                for (int g = 0; g < KS / 4; ++g) {
                    if (g + 1 < KS / 4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) (g & 1 ? b0 : b1)[j] = h_rd[64 * (4 * (g + 1) + j) + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bh = (g & 1 ? b1 : b0)[j];
                        az = __builtin_amdgcn_mfma_f32_16x16x4f32(w_rec[4 * g + j][0], bh, az, 0, 0, 0);
                        ar = __builtin_amdgcn_mfma_f32_16x16x4f32(w_rec[4 * g + j][1], bh, ar, 0, 0, 0);
                        an = __builtin_amdgcn_mfma_f32_16x16x4f32(w_rec[4 * g + j][2], bh, an, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gz = sigmoid_pre(az[e]), gr = sigmoid_pre(ar[e]);
                    const float nn = tanh_exp_pre(__builtin_fmaf(gr, an[e], ax[e]));      // (the record's candidate rows carry 2 log2 e)
                    const float hn = __builtin_fmaf(gz, hreg[e] - nn, nn);
                    hreg[e] = hn;
                    h_wr[(16 * wave + 4 * q + e) * NS + c] = hn;
                }
                par ^= 1;
            }
            __syncthreads();                                // the tick's barrier
        }
        __syncthreads();                                    // (4) the helpers have stored the rows
    }
#ifdef AIDAX_LP_TRACE
    if ((int)blockIdx.x == ((AIDAX_TUNE(a) >> 16) & 0xff)) __syncthreads();      // (lp_helper's extra barrier of the measurement build)
#endif
    __syncthreads();                                        // (5)
    if (valid && livef[c] != 0.f) {
        float* dst = a.nn + (size_t)sg * a.nn_stride + L.state_off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 16 * wave + 4 * q + e;
            if (u < Ht) dst[u] = hreg[e];
        }
    }
}


// ================================================================ k_gru_gs: k_gru_gm with the recurrent product on the bf16 matrix pipe
// What round 4 measured (scratch/uoverlap2.hip, uoverlap3.hip, profiles/r04_overlap.txt): v_mfma_f32_16x16x4_f32 runs at the
// VECTOR rate (32 cycles for 1024 MACs) and nothing else of the SIMD issues beside it — a VALU instruction behind an fp32 MFMA
// costs its full issue time whatever accumulator it touches, one wave or two; v_mfma_f32_16x16x32_bf16 does 8192 MACs in ~17
// cycles and hides up to six VALU instructions in its shadow. So the recurrent product U . h(t-1) is computed here from
// operands split EXACTLY into three bf16 terms each (w = w0 + w1 + w2, h = h0 + h1 + h2; 8 + 8 + 8 significant bits,
// round-to-nearest remainders): every bf16 x bf16 product is exact in fp32, the accumulation is fp32 as before, and of the
// nine term products NPROD are issued, smallest first — 9: the fp32 product to the last bit of its operands; 6: without
// w1 h2, w2 h1, w2 h2 (together <= 2^-23 |w h|: below the rounding of the fp32 accumulation that follows). Per wave and
// frame: 3 gates x 2 k-steps x NPROD bf16 MFMAs (36 x 17 cycles for NPROD = 6) instead of 48 fp32 MFMAs x 32.
// The weights arrive split from the packer (pack_mfma, gs record); a lane splits its own four h values after the cell update
// (v_cvt_pk_bf16_f32 rounds to nearest even; the remainders are exact) and writes them where the k-steps read them: h lives
// in LDS as B fragments [parity][term][k-step][lane][8 bf16], one ds_read_b128 per (term, k-step). The model inputs stay one
// fp32 k-step (x, PARAM1, PARAM2 are not products of h), Dense(H,1) is four FMAs on the lane's own h and two permlane swaps.
// Everything else — tiles, helper waves, barriers, the per-frame order of work — is k_gru_gm's.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t gs_lds_floats(int hidden, int n_frames, int n_helpers)
{
    const size_t nP = (size_t)(((n_frames < kLpChunk ? n_frames : kLpChunk) + 3) & ~3);
    const size_t ks2 = (size_t)(hidden + 31) / 32;
    return (size_t)kMfmaStreams * nP + 2 * 64 + 2 * 3 * ks2 * 64 * 4 /* h fragments */ + (size_t)((hidden + 1 + 3) & ~3) + kMfmaStreams
         + 2 * 8 * kMfmaStreams + (size_t)n_helpers * 2 * kChainHandFloats
#ifdef AIDAX_LP_TRACE
         + 2048
#endif
         ;
}

// a, b -> (bf16(a) | bf16(b) << 16), round to nearest even
__device__ __forceinline__ unsigned gs_pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = { static_cast<__bf16>(a), static_cast<__bf16>(b) };
    unsigned u = __builtin_bit_cast(unsigned, v);
    asm volatile("" : "+v"(u));                               // ONE v_cvt_pk_bf16_f32: left to itself the compiler converts each value again to widen it back
    return u;
}
// four fp32 values -> three terms of four bf16 each (term t: two dwords), v = t0 + t1 + t2 exactly
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gs_split4(const float (&v)[4], u32x2 (&t)[3])
{
#ifndef AIDAX_GS_NOPK
    // The residuals' subtractions two values per instruction (v_pk_add_f32: the same IEEE subtraction per value). The main waves of the
    // tick kernels share their SIMD with one helper wave: they issue like a lone wave, for which a packed instruction costs what a plain
    // one does (profiles/r05_lone_wave_issue.txt) — and this split sits between a tick's last MFMA and its barrier.
    gs_f32x2 r01 = { v[0], v[1] }, r23 = { v[2], v[3] };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned p01 = gs_pack_bf16(r01.x, r01.y), p23 = gs_pack_bf16(r23.x, r23.y);
        t[k] = u32x2{ p01, p23 };
        if (k < 2) {
            r01 = r01 - gs_f32x2{ __builtin_bit_cast(float, p01 << 16), __builtin_bit_cast(float, p01 & 0xffff0000u) };
            r23 = r23 - gs_f32x2{ __builtin_bit_cast(float, p23 << 16), __builtin_bit_cast(float, p23 & 0xffff0000u) };
        }
    }
    return;
#endif
    float r[4] = { v[0], v[1], v[2], v[3] };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned p01 = gs_pack_bf16(r[0], r[1]), p23 = gs_pack_bf16(r[2], r[3]);
        t[k] = u32x2{ p01, p23 };
        if (k < 2) {
            r[0] -= __builtin_bit_cast(float, p01 << 16);
            r[1] -= __builtin_bit_cast(float, p01 & 0xffff0000u);
            r[2] -= __builtin_bit_cast(float, p23 << 16);
            r[3] -= __builtin_bit_cast(float, p23 & 0xffff0000u);
        }
    }
}

template <int UT, int NHELP, int NPROD>
__global__ __launch_bounds__((UT + NHELP) * kWave) void k_gru_gs(LaunchArgs a, MfmaDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * UT, NW = UT, NS = kMfmaStreams, NT = NW * kWave;
    constexpr int KS2 = (H + 31) / 32;                      // bf16 k-steps (32 columns each; GRU-48: the second one half empty)
    constexpr int kFrag = 3 * KS2 * 64;                     // 16-byte B fragments of one parity: [term][k-step][lane]
    static_assert(NPROD == 6 || NPROD == 9, "six products (to fp32 rounding) or all nine");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int grp = (int)blockIdx.x;
    const int s_base = grp * NS;
    const int Ht = d.hidden_true;
    const int chunk = n < kLpChunk ? n : kLpChunk;
    const int nP = (chunk + 3) & ~3;
    float* xb    = smem;                                    // [NS][nP] audio rows
    float* xin   = xb + NS * nP;                            // [2][4][NS] model inputs of a frame
    u32x4* hB    = reinterpret_cast<u32x4*>(xin + 2 * 64);  // [2][3][KS2][64] h(t-1) as B fragments of the bf16 k-steps
    float* wdl   = reinterpret_cast<float*>(hB + 2 * kFrag);// Dense weights, bias at [H]
    float* livef = wdl + ((H + 1 + 3) & ~3);                // [NS]
    float* dpart = livef + NS;                              // [2][8][NS] Dense partial sums of the waves
    float* hands = dpart + 2 * 8 * NS;
#ifdef AIDAX_LP_TRACE
    // measurement build (scratch/gs_trace.py): the selected workgroup stamps the shader clock at six points of ticks 96..103 on
    // every wave (the helpers in lp_helper) and leaves the stamps in the first floats of its output rows
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(hands + NHELP * 2 * kChainHandFloats);
    constexpr int kTraceT0 = 96;
#endif
    if (wave >= NW) {
        lp_helper<H, NW, NHELP>(a, xb, xin, wdl, livef, dpart, hands, grp, nP);
        return;
    }
    const float* W = a.wpack;
    const MfmaLayer& L = d.L[0];
    const int q = lane >> 4, c = lane & 15;
    for (int i = tid; i < H + 1; i += NT) wdl[i] = W[d.wd_off + i];
    // the fp32 pieces of the gate-major record: input k-step and bias rows
    constexpr int KS = H / 4;
    const float* rec = W + d.gm_off + (size_t)wave * kWave * (3 + 3 * KS + 16);
    float w_in[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) w_in[g] = rec[g * kWave + lane];
    f32x4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(rec + (3 + 3 * KS) * kWave + (g * kWave + lane) * 4);
    // the recurrent weights as split bf16 A fragments: [gate][k-step][term]
    bf16x8 wq[3][KS2][3];
    {
        const u32x4* sp = reinterpret_cast<const u32x4*>(W + d.gs_off) + (size_t)wave * 3 * KS2 * 3 * kWave + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t) wq[g][ks][t] = __builtin_bit_cast(bf16x8, sp[((g * KS2 + ks) * 3 + t) * kWave]);
    }
    float dw[4];                                            // Dense weights of this lane's four units
    // h(t-1) of this lane's four units (16 wave + 4q + e) of stream s_base + c: registers for the launch
    const int sg = s_base + c;
    const bool valid = sg < (int)a.n_streams;
    const float* stp = a.nn + (size_t)(valid ? sg : 0) * a.nn_stride + L.state_off;
    float hreg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int u = 16 * wave + 4 * q + e;
        hreg[e] = (valid && u < Ht) ? stp[u] : 0.f;         // padded units rest at 0
        dw[e] = W[d.wd_off + u];                            // (zero for padded units)
    }
    // where this lane's four values sit in the fragments: k = 16 wave + 4q + e -> k-step wave / 2, lane row 2 (wave & 1) + q / 2,
    // elements 4 (q & 1) .. + 3 of that lane's eight: the upper or lower 8 bytes of its 16
    const int my_slot = (wave >> 1) * 64 + (2 * (wave & 1) + (q >> 1)) * 16 + c;
    const int my_half = q & 1;
    auto publish = [&](int parity) {
        if (AIDAX_TUNE(a) & 1048576) {
            // (bit 1048576, test build: term 0 of h only — an upper bound on what taking terms 1 and 2 of the split off the tick's critical
            // path could buy, the round-5 review's item 8; wrong output)
            u32x2 t0 = { gs_pack_bf16(hreg[0], hreg[1]), gs_pack_bf16(hreg[2], hreg[3]) };
            reinterpret_cast<u32x2*>(hB + parity * kFrag + my_slot)[my_half] = t0;
            return;
        }
        u32x2 t[3];
        gs_split4(hreg, t);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            reinterpret_cast<u32x2*>(hB + parity * kFrag + k * KS2 * 64 + my_slot)[my_half] = t[k];
    };
    if constexpr (H % 32 != 0) {                            // GRU-48: columns 48..63 of the second k-step (lane rows 2, 3) are nobody's: zero, for good
        for (int i = tid; i < 2 * 3 * 32; i += NT)
            hB[(i / 32) * KS2 * 64 + (KS2 - 1) * 64 + 32 + (i & 31)] = u32x4{ 0u, 0u, 0u, 0u };
    }
    publish(0);
    __syncthreads();                                        // (1)

    int par = 0;
    for (int base = 0; base < n; base += kLpChunk) {
        const int cnt = n - base < kLpChunk ? n - base : kLpChunk;
        for (int sl = wave; sl < NS; sl += NW) {            // every valid row: the chains run on net-off streams too
            const int s2 = s_base + sl;
            const bool lv = s2 < (int)a.n_streams;
            float* row = xb + sl * nP;
            const float* src = a.in + (size_t)(lv ? s2 : 0) * n + base;
            if (lv && ((n | base) & 3) == 0) load_block(row, src, cnt, lane);
            else for (int t = lane; t < cnt; t += kWave) row[t] = lv ? src[t] : 0.f;
        }
        __syncthreads();                                    // (2)
        __syncthreads();                                    // (3) the helpers have run the head of the pre pass and written frame 0's inputs
        const int ticks = cnt + 2;
        for (int tick = 0; tick < ticks; ++tick) {
            LP_STAMP(0);
            // Dense of frame tick-1 from the lane's own h(tick-1), in nine steps (four multiply-adds over the lane's units in a
            // fixed order, two permlane swaps over the lane rows, the store) — issued one behind every second z / r MFMA of
            // the frame loop: bf16 MFMAs leave the VALU free while they run (profiles/r04_overlap.txt)
            float dy = 0.f;
            Pair dp = { 0.f, 0.f };
            const bool dense_on = tick >= 1 && tick <= cnt;
            auto dense_step = [&](int k) {
                switch (k) {
                case 0: dy = dw[0] * hreg[0]; break;
                case 1: dy = __builtin_fmaf(dw[1], hreg[1], dy); break;
                case 2: dy = __builtin_fmaf(dw[2], hreg[2], dy); break;
                case 3: dy = __builtin_fmaf(dw[3], hreg[3], dy); break;
                case 4: dp = share_rows(dy); break;           // rows q and q ^ 1
                case 5: dy = dp.lo + dp.hi; break;
                case 6: dp = share_halves(dy); break;         // ... and the other half of the wave
                case 7: dy = dp.lo + dp.hi; break;
                case 8: if (dense_on && lane < NS) dpart[((tick & 1) * NW + wave) * NS + lane] = dy; break;
                default: break;
                }
            };
            if (tick < cnt) {
                const u32x4* h_rd = hB + par * kFrag + lane;
                // term products, smallest first: (weight term, h term); the fragments are requested in the order of their use
                constexpr int P9[9][2] = { {2, 2}, {1, 2}, {2, 1}, {0, 2}, {1, 1}, {2, 0}, {0, 1}, {1, 0}, {0, 0} };
                bf16x8 hb[KS2][3];
#pragma unroll
                for (int t = 2; t >= 0; --t)
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) hb[ks][t] = __builtin_bit_cast(bf16x8, h_rd[(t * KS2 + ks) * 64]);
                f32x4 acc[3] = { bias[0], bias[1], bias[2] };
                f32x4 ax = bias[3];
                const float bx = xin[(tick & 1) * 64 + lane];
                LP_STAMP(1);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[0], bx, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[1], bx, acc[1], 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[2], bx, ax, 0, 0, 0);
                // z and r first, two accumulators in turn ...
#pragma unroll
                for (int j = 0; j < NPROD * KS2; ++j) {
                    const int pi = 9 - NPROD + j / KS2, ks = j % KS2;
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[1][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    dense_step(j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ... then the candidate's chain, with the sigmoids of z and r in its shadow: 2^v for the eight values behind
                // the first four MFMAs, 1 / (1 + .) behind the next eight
                float sg8[8];
#pragma unroll
                for (int j = 0; j < NPROD * KS2; ++j) {
                    const int pi = 9 - NPROD + j / KS2, ks = j % KS2;
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[2][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 4) {
                        sg8[2 * j] = __builtin_amdgcn_exp2f(acc[(2 * j) >> 2][(2 * j) & 3]);
                        sg8[2 * j + 1] = __builtin_amdgcn_exp2f(acc[(2 * j + 1) >> 2][(2 * j + 1) & 3]);
                    } else if (j < 12) {
                        sg8[j - 4] = __builtin_amdgcn_rcpf(1.0f + sg8[j - 4]);      // sigmoid_pre: the rows carry -log2 e
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                LP_STAMP(2);
#ifdef AIDAX_GS_PRIO                                             // (measurement: the main wave's tail — cell update, split, publish — ahead of the helpers)
                __builtin_amdgcn_s_setprio(3);
#endif
#ifndef AIDAX_GS_NOPK
                // the cell update two units per instruction (v_pk_fma_f32 / v_pk_add_f32 around the two exponentials and reciprocals: the
                // same operations per unit, the same bits): 18 instructions for 28 between the last MFMA and the publish
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const gs_f32x2 gz = { sg8[e], sg8[e + 1] }, gr = { sg8[4 + e], sg8[5 + e] };
                    const gs_f32x2 pre = __builtin_elementwise_fma(gr, gs_f32x2{ acc[2][e], acc[2][e + 1] }, gs_f32x2{ ax[e], ax[e + 1] });      // (the record's candidate rows carry 2 log2 e)
                    const gs_f32x2 ex = { __builtin_amdgcn_exp2f(pre.x), __builtin_amdgcn_exp2f(pre.y) };
                    const gs_f32x2 sm = ex + gs_f32x2{ 1.0f, 1.0f };
                    const gs_f32x2 rc = { __builtin_amdgcn_rcpf(sm.x), __builtin_amdgcn_rcpf(sm.y) };
                    const gs_f32x2 nn = __builtin_elementwise_fma(gs_f32x2{ -2.0f, -2.0f }, rc, gs_f32x2{ 1.0f, 1.0f });     // tanh_exp_pre
                    const gs_f32x2 hh = __builtin_elementwise_fma(gz, gs_f32x2{ hreg[e], hreg[e + 1] } - nn, nn);
                    hreg[e] = hh.x; hreg[e + 1] = hh.y;
                }
#else
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gz = sg8[e], gr = sg8[4 + e];
                    const float nn = tanh_exp_pre(__builtin_fmaf(gr, acc[2][e], ax[e]));      // (the record's candidate rows carry 2 log2 e)
                    hreg[e] = __builtin_fmaf(gz, hreg[e] - nn, nn);
                }
#endif
                LP_STAMP(3);
                par ^= 1;
                publish(par);
                LP_STAMP(4);
#ifdef AIDAX_GS_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            } else if (dense_on) {
#pragma unroll
                for (int k = 0; k < 9; ++k) dense_step(k);
            }
            __syncthreads();                                // the tick's barrier
            LP_STAMP(5);
        }
        __syncthreads();                                    // (4) the helpers have stored the rows
    }
#ifdef AIDAX_LP_TRACE
    if ((int)blockIdx.x == ((AIDAX_TUNE(a) >> 16) & 0xff)) __syncthreads();
#endif
    __syncthreads();                                        // (5)
#ifdef AIDAX_LP_TRACE
    if ((int)blockIdx.x == ((AIDAX_TUNE(a) >> 16) & 0xff))
        for (int i = tid; i < 2 * 768; i += NT) a.out[(size_t)s_base * n + i] = reinterpret_cast<const float*>(trace)[i];
#endif
    if (valid && livef[c] != 0.f) {
        float* dst = a.nn + (size_t)sg * a.nn_stride + L.state_off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 16 * wave + 4 * q + e;
            if (u < Ht) dst[u] = hreg[e];
        }
    }
}


// ================================================================ k_lstm_gs: k_gru_gs's structure for one-layer LSTMs
// UT main waves, one per 16 units (the four gate tiles i | f | g | o of those units: a lane's accumulator quads are the four gates of
// four ADJACENT units of one stream), NHELP helper waves with the whole DSP chain, the model inputs and the Dense's tail
// (lp_helper) — the whole run() in one launch. Per frame and main wave: 4 fp32 MFMAs (the model inputs' k-step), 4 x KS2 x NPROD
// bf16 MFMAs (48 for 64 units) — i and g first, two accumulators in turn with the Dense's steps behind them, then f and o with
// sigmoid(i), tanh(g) and their product dealt out in their shadow (bf16 MFMAs leave the VALU free, profiles/r04_overlap.txt) —
// then the rest of the cell update (the same operations in the same order as k_mfma_ls's cell_step: sigmoid_pre, tanh_rat),
// the split of the four new h values (gs_split4) and three 8-byte fragment writes, one barrier. The record: pack_mfma, gs_off.
template <int UT, int NHELP, int NPROD>
__global__ __launch_bounds__((UT + NHELP) * kWave) void k_lstm_gs(LaunchArgs a, MfmaDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * UT, NW = UT, NS = kMfmaStreams, NT = NW * kWave;
    constexpr int KS2 = (H + 31) / 32;
    constexpr int kFrag = 3 * KS2 * 64;
    static_assert(NPROD == 6 || NPROD == 9, "six products (to fp32 rounding) or all nine");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int grp = (int)blockIdx.x;
    const int s_base = grp * NS;
    const int Ht = d.hidden_true;
    const int chunk = n < kLpChunk ? n : kLpChunk;
    const int nP = (chunk + 3) & ~3;
    float* xb    = smem;                                    // [NS][nP] audio rows
    float* xin   = xb + NS * nP;                            // [2][4][NS] model inputs of a frame
    u32x4* hB    = reinterpret_cast<u32x4*>(xin + 2 * 64);  // [2][3][KS2][64] h(t-1) as B fragments of the bf16 k-steps
    float* wdl   = reinterpret_cast<float*>(hB + 2 * kFrag);// Dense weights, bias at [H]
    float* livef = wdl + ((H + 1 + 3) & ~3);                // [NS]
    float* dpart = livef + NS;                              // [2][8][NS] Dense partial sums of the waves
    float* hands = dpart + 2 * 8 * NS;
    if (wave >= NW) {
        lp_helper<H, NW, NHELP>(a, xb, xin, wdl, livef, dpart, hands, grp, nP);
        return;
    }
    const float* W = a.wpack;
    const MfmaLayer& L = d.L[0];
    const int q = lane >> 4, c = lane & 15;
    for (int i = tid; i < H + 1; i += NT) wdl[i] = W[d.wd_off + i];
    constexpr size_t kRecFloats = 4 * kWave + 4 * kWave * 4 + (size_t)4 * KS2 * 3 * kWave * 4;      // per wave: input k-step, bias quads, split fragments
    const float* rec = W + d.gs_off + (size_t)wave * kRecFloats;
    float w_in[4];
    f32x4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        w_in[g] = rec[g * kWave + lane];
        bias[g] = *reinterpret_cast<const f32x4*>(rec + 4 * kWave + (g * kWave + lane) * 4);
    }
    bf16x8 wq[4][KS2][3];
    {
        const u32x4* sp = reinterpret_cast<const u32x4*>(rec + 4 * kWave + 4 * kWave * 4) + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t) wq[g][ks][t] = __builtin_bit_cast(bf16x8, sp[((g * KS2 + ks) * 3 + t) * kWave]);
    }
    // h(t-1) and c(t-1) of this lane's four units (16 wave + 4q + e) of stream s_base + c: registers for the launch
    const int sg = s_base + c;
    const bool valid = sg < (int)a.n_streams;
    const float* stp = a.nn + (size_t)(valid ? sg : 0) * a.nn_stride + L.state_off;
    float hreg[4], creg[4], dw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int u = 16 * wave + 4 * q + e;
        hreg[e] = (valid && u < Ht) ? stp[u] : 0.f;         // padded units rest at 0
        creg[e] = (valid && u < Ht) ? stp[Ht + u] : 0.f;
        dw[e] = W[d.wd_off + u];                            // (zero for padded units)
    }
    const int my_slot = (wave >> 1) * 64 + (2 * (wave & 1) + (q >> 1)) * 16 + c;      // (k_gru_gs: where a lane's four values sit in the fragments)
    const int my_half = q & 1;
    auto publish = [&](int parity) {
        u32x2 t[3];
        gs_split4(hreg, t);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            reinterpret_cast<u32x2*>(hB + parity * kFrag + k * KS2 * 64 + my_slot)[my_half] = t[k];
    };
    if constexpr (H % 32 != 0) {                            // columns H .. 32 KS2 - 1 of the last k-step are nobody's: zero, for good
        for (int i = tid; i < 2 * 3 * 32; i += NT)
            hB[(i / 32) * KS2 * 64 + (KS2 - 1) * 64 + 32 + (i & 31)] = u32x4{ 0u, 0u, 0u, 0u };
    }
    publish(0);
    __syncthreads();                                        // (1)

    int par = 0;
    for (int base = 0; base < n; base += kLpChunk) {
        const int cnt = n - base < kLpChunk ? n - base : kLpChunk;
        for (int sl = wave; sl < NS; sl += NW) {            // every valid row: the chains run on net-off streams too
            const int s2 = s_base + sl;
            const bool lv = s2 < (int)a.n_streams;
            float* row = xb + sl * nP;
            const float* src = a.in + (size_t)(lv ? s2 : 0) * n + base;
            if (lv && ((n | base) & 3) == 0) load_block(row, src, cnt, lane);
            else for (int t = lane; t < cnt; t += kWave) row[t] = lv ? src[t] : 0.f;
        }
        __syncthreads();                                    // (2)
        __syncthreads();                                    // (3) the helpers have run the head of the pre pass and written frame 0's inputs
        const int ticks = cnt + 2;
        for (int tick = 0; tick < ticks; ++tick) {
            float dy = 0.f;
            Pair dp = { 0.f, 0.f };
            const bool dense_on = tick >= 1 && tick <= cnt;
            auto dense_step = [&](int k) {                  // (k_gru_gs: the Dense of frame tick-1 from the lane's own h(tick-1), nine steps)
                switch (k) {
                case 0: dy = dw[0] * hreg[0]; break;
                case 1: dy = __builtin_fmaf(dw[1], hreg[1], dy); break;
                case 2: dy = __builtin_fmaf(dw[2], hreg[2], dy); break;
                case 3: dy = __builtin_fmaf(dw[3], hreg[3], dy); break;
                case 4: dp = share_rows(dy); break;
                case 5: dy = dp.lo + dp.hi; break;
                case 6: dp = share_halves(dy); break;
                case 7: dy = dp.lo + dp.hi; break;
                case 8: if (dense_on && lane < NS) dpart[((tick & 1) * NW + wave) * NS + lane] = dy; break;
                default: break;
                }
            };
            if (tick < cnt) {
                const u32x4* h_rd = hB + par * kFrag + lane;
                constexpr int P9[9][2] = { {2, 2}, {1, 2}, {2, 1}, {0, 2}, {1, 1}, {2, 0}, {0, 1}, {1, 0}, {0, 0} };      // (weight term, h term), smallest first
                bf16x8 hb[KS2][3];
#pragma unroll
                for (int t = 2; t >= 0; --t)
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) hb[ks][t] = __builtin_bit_cast(bf16x8, h_rd[(t * KS2 + ks) * 64]);
                f32x4 acc[4] = { bias[0], bias[1], bias[2], bias[3] };
                const float bx = xin[(tick & 1) * 64 + lane];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[g], bx, acc[g], 0, 0, 0);
                // i and g first, two accumulators in turn, the Dense's steps behind them ...
#pragma unroll
                for (int j = 0; j < NPROD * KS2; ++j) {
                    const int pi = 9 - NPROD + j / KS2, ks = j % KS2;
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[0][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[0], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[2][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    dense_step(j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ... then f and o, with sigmoid(i) * tanh(g) of the four units in their shadow: six steps per unit (sigmoid_pre: the
                // sigmoid rows carry -log2 e; tanh_rat: the odd rational x P(x^2) / Q(x^2))
                float ig[4], tx[4], tu[4], tp[4], tq[4], te[4];
                auto ig_step = [&](int e, int k) {
                    switch (k) {
                    case 0: tx[e] = tanh_rat_clamp(acc[2][e]); tu[e] = tx[e] * tx[e]; te[e] = __builtin_amdgcn_exp2f(acc[0][e]); break;
                    case 1: tp[e] = __builtin_fmaf(kTanhP[6], tu[e], kTanhP[5]); tp[e] = __builtin_fmaf(tp[e], tu[e], kTanhP[4]); tp[e] = __builtin_fmaf(tp[e], tu[e], kTanhP[3]); break;
                    case 2: tp[e] = __builtin_fmaf(tp[e], tu[e], kTanhP[2]); tp[e] = __builtin_fmaf(tp[e], tu[e], kTanhP[1]); tp[e] = __builtin_fmaf(tp[e], tu[e], kTanhP[0]); break;
                    case 3: tq[e] = __builtin_fmaf(kTanhQ[3], tu[e], kTanhQ[2]); tq[e] = __builtin_fmaf(tq[e], tu[e], kTanhQ[1]); tq[e] = __builtin_fmaf(tq[e], tu[e], kTanhQ[0]); break;
                    case 4: te[e] = __builtin_amdgcn_rcpf(1.0f + te[e]); tq[e] = __builtin_amdgcn_rcpf(tq[e]); break;
                    case 5: ig[e] = te[e] * ((tp[e] * tx[e]) * tq[e]); break;
                    default: break;
                    }
                };
#pragma unroll
                for (int j = 0; j < NPROD * KS2; ++j) {
                    const int pi = 9 - NPROD + j / KS2, ks = j % KS2;
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[1][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[1], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[3][ks][P9[pi][0]], hb[ks][P9[pi][1]], acc[3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < 12) { ig_step((2 * j) & 3, (2 * j) >> 2); ig_step((2 * j + 1) & 3, (2 * j + 1) >> 2); }      // 24 steps: the units advance together
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1][e]));
                    const float o = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3][e]));
                    creg[e] = __builtin_fmaf(f, creg[e], ig[e]);
                    hreg[e] = o * tanh_rat(creg[e]);
                }
                par ^= 1;
                publish(par);
            } else if (dense_on) {
#pragma unroll
                for (int k = 0; k < 9; ++k) dense_step(k);
            }
            __syncthreads();                                // the tick's barrier
        }
        __syncthreads();                                    // (4) the helpers have stored the rows
    }
    __syncthreads();                                        // (5)
    if (valid && livef[c] != 0.f) {
        float* dst = a.nn + (size_t)sg * a.nn_stride + L.state_off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 16 * wave + 4 * q + e;
            if (u < Ht) { dst[u] = hreg[e]; dst[Ht + u] = creg[e]; }
        }
    }
}


// ================================================================ k_mfma_ls: k_mfma_lp with the contractions on the bf16 matrix pipe
// The layer-pipelined kernel above (one workgroup per (16 streams, layer), the layer's fragments resident in registers, h on
// its way up through a ring in global memory, every wait bounded) with every fp32 product W . h computed as NPROD bf16 term
// products of operands split exactly into three bf16 terms — see k_gru_gs for the arithmetic and why (profiles/r04_overlap.txt:
// the fp32 MFMAs run at the vector rate and stall the VALU; the bf16 ones are sixteen times denser and hide it).
//   * Weights: the packer's ls record, [wave][tile][segment][k-step of 32][term] fragments of 8 bf16 per lane. A wave keeps
//     only the tile segments it multiplies with: its TPW own-h segments plus, on the first layer, the input-half segments of
//     the mw tiles it starts for the layer above — on the others, the input-half segments of the TPW - mw tiles that do not
//     arrive started (LSTM-96 x 2: five segments of 36 registers).
//   * h(t-1) of a layer lives in LDS as B fragments [parity][term][k-step][lane][8 bf16]; a lane splits its own h values after
//     the cell update and writes the three 2-byte terms where the k-steps read them. A frame in the ring is the fragments
//     as they lie in LDS (plus the started tiles, as before): the layer above copies them in and multiplies.
//   * Dense(H,1): multiply-adds on the lane's own h and two permlane swaps, the waves' partial sums added a tick later as before.
//   * The first layer's model inputs stay one fp32 k-step.
// Hand-over protocol, counters, give-up, the chain passes around the body: k_mfma_lp's, line for line.
// Geometry: k_mfma_lp's eight (four) waves while a wave's resident fragments fit 150 of its 256 registers; wider layers run FOUR
// waves — one per SIMD, 512 registers each (the fragments are MFMA operands only and may live in AGPRs) — with half of the
// upper layer's tiles started below, so that both workgroups of a group issue the same number of MFMAs (LSTM-96 x 2: six
// tiles per wave, nine resident segments of 36 registers, 162 bf16 MFMAs per wave and frame in either layer).
struct LsGeo { int nw, tpw, m, nx; };      // nx: resident tile segments of a wave besides its own-h ones (the larger role's; none for a lone layer)
__host__ __device__ constexpr int ls_ks2(int hidden) { return (hidden + 31) / 32; }
__host__ __device__ constexpr int ls_frag_vecs(int hidden) { return 3 * ls_ks2(hidden) * 64; }      // 16-byte vectors of one h buffer
__host__ __device__ constexpr int ls_segments(int tpw, int m) { return tpw + (tpw - m > m ? tpw - m : m); }      // resident tile segments of a wave (the larger role)
__host__ __device__ constexpr LsGeo ls_geo(int n_layers, int hidden)
{
    if (hidden < 16 || hidden % 16 != 0 || n_layers < 1) return LsGeo{ 0, 0, 0, 0 };
#ifndef AIDAX_LS_M96
#define AIDAX_LS_M96 2      // LSTM-96 x 2 on eight waves: tiles per wave the lower layer starts for the upper one (1: 32 / 40 tile segments per workgroup, 2: 40 / 32)
#endif
    // registers a wave spends on resident fragments: all three terms of its own-h segments, term 0 of the others
    const int nw = mfma_waves(hidden), tpw = hidden / 4 / nw, m = lp_moved_tiles(n_layers, tpw, nw) == 2 ? AIDAX_LS_M96 : lp_moved_tiles(n_layers, tpw, nw);
    const int nx = n_layers == 1 ? 0 : ls_segments(tpw, m) - tpw;
    if (tpw * ls_ks2(hidden) * 12 + nx * ls_ks2(hidden) * 4 <= 140) return LsGeo{ nw, tpw, m, nx };
    const int tpw4 = hidden / 16, m4 = n_layers == 2 ? tpw4 / 2 : 0, nx4 = n_layers == 1 ? 0 : ls_segments(tpw4, m4) - tpw4;
    if (n_layers <= 2 && tpw4 * ls_ks2(hidden) * 12 + nx4 * ls_ks2(hidden) * 4 <= 330) return LsGeo{ 4, tpw4, m4, nx4 };
    return LsGeo{ 0, 0, 0, 0 };
}
__host__ __device__ inline size_t ls_lds_floats(int hidden, int n_frames, LsGeo g)
{
    const size_t nP = (size_t)(((n_frames < kLpChunk ? n_frames : kLpChunk) + 3) & ~3);
    return (size_t)kMfmaStreams * nP                          /* xb: audio rows (first and last layer)                     */
         + 2 * 64                                             /* xin[parity][4][n]  (first layer)                          */
         + (size_t)4 * ls_frag_vecs(hidden) * 4               /* below[parity], hT[parity]: B fragments                     */
         + (size_t)2 * hidden * 4                             /* bias rows [unit][4]: own layer, layer above (moved tiles)  */
         + (size_t)((hidden + 1 + 3) & ~3)                    /* Dense weights + bias (last layer)                         */
         + kMfmaStreams                                       /* live flags                                                */
         + 2 * 8 * kMfmaStreams                               /* Dense partial sums [parity][wave][n]                      */
         + (size_t)g.nw * g.nx * ls_ks2(hidden) * 2 * 64 * 4;      /* second- and third-term fragments of the segments nobody waits for */
}
// one frame in the ring: the h fragments, then (M > 0) the started tiles [wave][tile][lane] x 4 gate rows
__host__ __device__ constexpr size_t ls_slot_floats(int hidden, int waves, int m) { return (size_t)ls_frag_vecs(hidden) * 4 + (size_t)waves * m * kWave * 4; }
__host__ __device__ constexpr size_t ls_ring_floats(int hidden, int waves, int m) { return (size_t)kLpRing * ls_slot_floats(hidden, waves, m); }

// acc[tile] += A(segment of the tile) . B for the tiles [TL0, TL1) of one segment kind. The B fragments of one h buffer are read
// into registers ONCE (ls_load_frags: nine 16-byte reads for LSTM-96) and serve every phase of the tick that multiplies with
// that buffer — read one by one between the MFMAs, the three MFMAs of a fragment's first product do not cover its LDS latency
// and the wave stalls once per fragment. Term products grouped by the h term, the large ones first: (w0 w1 w2) h0 | (w0 w1) h1 |
// w0 h2; NPROD = 9: all three weight terms for every h term. `wfrag(tl, ks, term)` supplies the A fragment of a tile;
// `valu(step)` is called behind every MFMA group (the tiles of one product and k-step) with a running index — the caller's
// VALU work that issues in the MFMAs' shadow.
template <int KS2> struct LsFrags { bf16x8 v[3][KS2]; };       // one h buffer's B fragments in registers: [term][k-step]
template <int KS2>
__device__ __forceinline__ void ls_load_frags(LsFrags<KS2>& hb, const u32x4* frags, int lane, int tune = 0)
{
    if (tune & 2097152) {
        // (bit 2097152, test build: ONE 16-byte read instead of 3 KS2 — an upper bound on what fewer LDS bytes per tick could buy
        // (the round-5 review's item 6 a: the eight waves of a CU each read all nine h fragments, 72 KiB per tick); wrong output)
        const bf16x8 v = __builtin_bit_cast(bf16x8, frags[lane]);
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) hb.v[th][ks] = v;
        return;
    }
#pragma unroll
    for (int th = 0; th < 3; ++th)                          // (term 0 first: it meets the most weight terms and is multiplied first)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) hb.v[th][ks] = __builtin_bit_cast(bf16x8, frags[(th * KS2 + ks) * 64 + lane]);
}
template <int KS2, int NPROD, int TL0, int TL1, int NACC, typename WFrag, typename Valu>
__device__ __forceinline__ void ls_gates(f32x4 (&acc)[NACC], const LsFrags<KS2>& hb, WFrag wfrag, Valu valu)
{
    if constexpr (TL0 < TL1) {
        int step = 0;
#pragma unroll
        for (int th = 0; th < 3; ++th) {                    // h term 0, 1, 2: (w0 w1 w2) h0 | (w0 w1) h1 | w0 h2
            const int n_w = NPROD == 9 ? 3 : 3 - th;        // weight terms this h term meets: w0 .. w(n_w - 1)
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int tw = 0; tw < n_w; ++tw) {
#pragma unroll
                    for (int tl = TL0; tl < TL1; ++tl)
#ifdef AIDAX_LS_NOMFMA                                            // (measurement build: everything but the matrix instructions)
                        acc[tl][0] += __builtin_bit_cast(f32x4, hb.v[th][ks])[0] * __builtin_bit_cast(f32x4, wfrag(tl, ks, tw))[0];
#else
                        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag(tl, ks, tw), hb.v[th][ks], acc[tl], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    valu(step++);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
}
// MFMA groups ls_gates issues (= calls of `valu`) per call
__host__ __device__ constexpr int ls_gate_groups(int ks2, int nprod) { return nprod * ks2; }

#ifdef AIDAX_LP_TRACE
// measurement build (scratch/ls_trace.py): the workgroups of stream group (tune >> 16 & 0xff) stamp the shader clock at eight points
// of ticks 96..103 on every wave, straight into the pinned fault buffer: [layer is last][tick][wave][stamp] u64 behind word 16
#define LS_STAMP(k) do { if (grp == ((AIDAX_TUNE(a) >> 16) & 0xff) && tick >= 96 && tick < 104) {                                       \
        __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); __builtin_amdgcn_sched_barrier(0);            \
        if (lane == 0) reinterpret_cast<unsigned long long*>(fault + 16)[(((last ? 1 : 0) * 8 + (tick - 96)) * 4 + (wave & 3)) * 8 + (k)] = t_; } } while (0)
#else
#define LS_STAMP(k) do {} while (0)
#endif

// M: tiles per wave the first layer starts for the layer above / that arrive started; CELL: 0 LSTM, 1 GRU (this layer's)
template <int TPW, int NW, int M, bool FIRST, bool LAST, bool CHAIN, int NPROD, int CELL>
__device__ __forceinline__ void ls_body(const LaunchArgs& a, const MfmaDesc& d, float* ring, uint32_t* counters, uint32_t* fault,
                                        float* smem, int grp, int l)
{
    constexpr int H = 4 * TPW * NW;
    constexpr int NT = NW * kWave;
    constexpr int NS = kMfmaStreams;
    constexpr int KS2 = ls_ks2(H);
    constexpr int kFrag = ls_frag_vecs(H);                  // 16-byte vectors of one h buffer
    constexpr int MW = M;
    constexpr int NSEG = FIRST ? TPW + MW : 2 * TPW - MW;   // resident tile segments of this wave
    constexpr int MA = MW > 0 ? MW : 1;
    constexpr size_t kSlot = ls_slot_floats(H, NW, M);      // floats of one ring frame
    constexpr bool chain = CHAIN, first = FIRST, last = LAST;
    static_assert(!(FIRST && LAST) || M == 0, "a lone layer starts no tiles for a layer above");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;
    const int n = (int)a.n_frames;
    const int NL = d.n_layers;
    const int Ht = d.hidden_true;
    const int I = a.input_size;
    const int blk = (int)blockIdx.x;
    const int s_base = grp * NS;
    const int chunk = n < kLpChunk ? n : kLpChunk;
    const int nP = (chunk + 3) & ~3;

    float* xb    = smem;                                    // [NS][nP]
    float* xin   = xb + NS * nP;                            // [2][4][NS]
    u32x4* below = reinterpret_cast<u32x4*>(xin + 2 * 64);  // [2][kFrag]  h of the layer below as B fragments
    u32x4* hT    = below + 2 * kFrag;                       // [2][kFrag]  own h(t-1) as B fragments
    float* hS    = reinterpret_cast<float*>(below);         // [H][NS]  h and c as fp32 on their way between the state records and the lanes:
    float* cT    = hS + H * NS;                             // [H][NS]  in `below`, before the first and after the last frame
    float* biasL = reinterpret_cast<float*>(hT + 2 * kFrag);// [H][4] own layer, then [H][4] of the layer above (first layer, MW > 0)
    float* wdl   = biasL + 2 * H * 4;                       // Dense weights, bias at [H]
    float* livef = wdl + ((H + 1 + 3) & ~3);                // [NS]
    float* dpart = livef + NS;                              // [2][NW][NS] Dense partial sums of the waves (last layer)

    const float* W = a.wpack;
    const MfmaLayer& L = d.L[l];

    // ---- per-stream bookkeeping: lanes tid < NS own stream s_base+tid (live flag; PARAM smoothers on the first layer)
    float p_mem[2] = { 0.f, 0.f }, p_tgt[2] = { 0.f, 0.f }, p_step[2] = { 0.f, 0.f };
    uint32_t pending = 0, st_pending0 = 0;
    bool mine_live = false;
    if (tid < NS) {
        const int sg = s_base + tid;
        const bool valid = sg < (int)a.n_streams;
        if (valid) {
            StreamState& st = a.st[sg];
            p_mem[0] = st.p_mem[0]; p_mem[1] = st.p_mem[1];
            p_tgt[0] = st.p_tgt[0]; p_tgt[1] = st.p_tgt[1];
            p_step[0] = st.p_step[0]; p_step[1] = st.p_step[1];
            pending = st_pending0 = st.pending;
            const StreamCtl& ctl = a.ctl[sg];
            const uint32_t flags = ctl.flags;
            mine_live = n != 0 && (flags & CTL_ENABLED) && (flags & CTL_NET_ON);      // :607-619, :631-632
            if (mine_live) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {                // LinearValueSmoother::setTargetValue (:209-216)
                    const float nt = ctl.p_target[i];
                    if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
                        p_tgt[i] = nt;
                        p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
                    }
                }
                if (pending & PEND_PARAM_FIRST) {            // paramFirstRun (:636-640)
                    pending &= ~PEND_PARAM_FIRST;
                    p_mem[0] = p_tgt[0];
                    p_mem[1] = p_tgt[1];
                }
            }
        }
        livef[tid] = mine_live ? 1.f : 0.f;
    }
    if (last) for (int i = tid; i < H + 1; i += NT) wdl[i] = W[d.wd_off + i];
    for (int i = tid; i < H * 4; i += NT) {
        biasL[i] = W[L.b_off + i];
        if (first && MW > 0) biasL[H * 4 + i] = W[d.L[1].b_off + i];
    }
    // this layer's recurrent state -> LDS as fp32; the fragment buffers start from zero (padded k-steps stay zero for good)
    for (int i = tid; i < H * NS; i += NT) {
        const int u = i / NS, sn = i % NS, sg = s_base + sn;
        const bool valid = sg < (int)a.n_streams;
        const float* stp = a.nn + (size_t)(valid ? sg : 0) * a.nn_stride + L.state_off;
        hS[i] = (valid && u < Ht) ? stp[u] : 0.f;           // padded units rest at 0
        cT[i] = (valid && u < Ht && L.cell == 0) ? stp[Ht + u] : 0.f;
    }
    for (int i = tid; i < 2 * kFrag; i += NT) hT[i] = u32x4{ 0u, 0u, 0u, 0u };

    // ---- resident fragments. The TPW own-h segments (every tick's critical path): all three terms in registers. The NX
    // others — the input half of the tiles started for the layer above (first layer) / of the tiles that start here (others),
    // MFMAs nobody waits for — keep term 0 in registers and terms 1 and 2 (three of the six products) in LDS, each wave its own.
    constexpr int NX = NSEG - TPW;
    constexpr int NXA = NX > 0 ? NX : 1;
    bf16x8 wq[TPW][KS2][3];
    bf16x8 wx[NXA][KS2];
    u32x4* w2x = reinterpret_cast<u32x4*>(dpart + 2 * 8 * NS) + (size_t)wave * NXA * KS2 * 2 * 64 + lane;      // [x][ks][term - 1] at stride 64
    {
        // the ls record: layer 0 holds one segment per tile (own h), the others two (h below | own h)
        const size_t per_seg = (size_t)KS2 * 3 * kWave;                     // u32x4 per (tile, segment)
        const u32x4* rec = reinterpret_cast<const u32x4*>(W + d.ls_off);
        auto seg_ptr = [&](int ll, int tl, int seg_in_rec) {
            return rec + (ll == 0 ? 0 : (size_t)NW * TPW * per_seg * (size_t)(1 + 2 * (ll - 1)))
                       + ((size_t)(wave * TPW + tl) * (ll == 0 ? 1 : 2) + seg_in_rec) * per_seg + lane;
        };
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {                                  // own h(t-1)
            const u32x4* sp = seg_ptr(l, tl, first ? 0 : 1);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t) wq[tl][ks][t] = __builtin_bit_cast(bf16x8, sp[(ks * 3 + t) * kWave]);
        }
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            // first layer: the layer above's input half of the tiles this wave starts; others: h below, for the tiles that start here
            const u32x4* sp = first ? seg_ptr(1, x, 0) : seg_ptr(l, MW + x, 0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                wx[x][ks] = __builtin_bit_cast(bf16x8, sp[(ks * 3 + 0) * kWave]);
                w2x[((x * KS2 + ks) * 2 + 0) * 64] = sp[(ks * 3 + 1) * kWave];
                w2x[((x * KS2 + ks) * 2 + 1) * 64] = sp[(ks * 3 + 2) * kWave];
            }
        }
    }
    float w_in0[TPW];                                       // (pack_mfma lays the input k-step out by ITS waves and tiles: d.waves x d.tpw)
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        const int T = wave * TPW + tl;
        w_in0[tl] = first ? W[d.L[0].w_in_off + ((size_t)(T / d.tpw) * kWave + lane) * d.tpw + T % d.tpw] : 0.f;
    }
    float wdu[TPW];                                         // Dense weights of this lane's units (last layer)
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) wdu[tl] = last ? W[d.wd_off + 4 * (wave * TPW + tl) + q] : 0.f;

    // ---- ring bookkeeping (k_mfma_lp's): counters count frames since the buffers were allocated and are equal on both sides
    // between launches, so a workgroup starts from its own side's value
    const size_t ring_stride = ls_ring_floats(H, NW, M);
    float* ring_out = last ? nullptr : ring + ((size_t)grp * (NL - 1) + l) * ring_stride;
    const float* ring_in = first ? nullptr : ring + ((size_t)grp * (NL - 1) + (l - 1)) * ring_stride;
    const size_t ring_bytes = ring_stride * sizeof(float);
    const __amdgpu_buffer_rsrc_t rs_out = lp_rsrc(last ? ring : ring_out, ring_bytes);
    const __amdgpu_buffer_rsrc_t rs_in = lp_rsrc(first ? ring : ring_in, ring_bytes);
    uint32_t* cnt_out = last ? nullptr : counters + ((size_t)grp * (NL - 1) + l) * kLpCounterStride;
    uint32_t* cnt_in = first ? nullptr : counters + ((size_t)grp * (NL - 1) + (l - 1)) * kLpCounterStride;
    auto give_up = [&]() { __hip_atomic_fetch_add(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    if ((AIDAX_TUNE(a) & 16) && blk == 0 && tid == 0) give_up();      // test hook: report a give-up that did not happen
    const uint32_t base_out = cnt_out ? __hip_atomic_load(cnt_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t base_in = cnt_in ? __hip_atomic_load(cnt_in + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t known_free = kLpRing;                         // frames of this launch the ring above is known to have room for
    uint32_t known_below = 0;                              // frames of this launch known to exist below
    __syncthreads();

    // ---- this lane's (unit, stream) pairs — one per tile: h and c in registers for the launch
    float hv[TPW], creg[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        hv[tl] = hS[(wave * TPW + tl) * 64 + lane];         // unit 4T + q, stream lane & 15
        creg[tl] = cT[(wave * TPW + tl) * 64 + lane];
    }
    __syncthreads();                                        // (the staging area becomes `below`)
    for (int i = tid; i < 2 * kFrag; i += NT) below[i] = u32x4{ 0u, 0u, 0u, 0u };
    // where a lane's h values go in the fragments: unit u = 4T + q -> k-step u / 32, lane row (u % 32) / 8, element u % 8
    auto publish_tile = [&](u32x4* dst, int tl) {
#ifdef AIDAX_LS_NOPUBLISH                                            // (measurement build, wrong results: what split + fragment writes cost)
        (void)dst; (void)tl;
        return;
#endif
        const int u = 4 * (wave * TPW + tl) + q;
        uint16_t* p16 = reinterpret_cast<uint16_t*>(dst + (u >> 5) * 64 + ((u & 31) >> 3) * 16 + (lane & 15)) + (u & 7);
        float r = hv[tl];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const __bf16 b = static_cast<__bf16>(r);         // round to nearest even
            p16[(size_t)t * KS2 * 64 * 8] = __builtin_bit_cast(uint16_t, b);
            if (t < 2) r -= static_cast<float>(b);           // exact
        }
    };
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) publish_tile(hT, tl);
    __syncthreads();

    // ---- the cell update of one tile in steps of three to five instructions, so that it can be dealt out behind the MFMA
    // groups of the NEXT tiles (bf16 MFMAs leave the VALU free while they run). Same operations in the same order as
    // sigmoid_pre / tanh_rat / tanh_exp_pre; the last step splits the new h and writes its fragments.
    struct CellTmp { float e0, e1, e2, x, u, p, qq, gg; };
    constexpr int NSTEP = CELL == 0 ? 12 : 5;
    CellTmp ct[TPW];
    auto cell_step = [&](int tl, int k, const f32x4& g, u32x4* h_dst) {
        CellTmp& t = ct[tl];
#ifdef AIDAX_LS_NOCELL                                               // (measurement build, wrong results: what the cell update costs)
        if (k == NSTEP - 1) { hv[tl] = g.x * 1e-9f; publish_tile(h_dst, tl); }
        return;
#endif
        if constexpr (CELL == 0) {                          // LSTM: g = (i, f, g, o), the sigmoid rows carrying -log2 e
            switch (k) {
            case 0: t.x = tanh_rat_clamp(g.z); t.u = t.x * t.x; t.e0 = __builtin_amdgcn_exp2f(g.x); break;
            case 1: t.p = __builtin_fmaf(kTanhP[6], t.u, kTanhP[5]); t.p = __builtin_fmaf(t.p, t.u, kTanhP[4]); t.e1 = __builtin_amdgcn_exp2f(g.y); break;
            case 2: t.p = __builtin_fmaf(t.p, t.u, kTanhP[3]); t.p = __builtin_fmaf(t.p, t.u, kTanhP[2]); t.e2 = __builtin_amdgcn_exp2f(g.w); break;
            case 3: t.p = __builtin_fmaf(t.p, t.u, kTanhP[1]); t.p = __builtin_fmaf(t.p, t.u, kTanhP[0]); t.qq = __builtin_fmaf(kTanhQ[3], t.u, kTanhQ[2]); break;
            case 4: t.qq = __builtin_fmaf(t.qq, t.u, kTanhQ[1]); t.qq = __builtin_fmaf(t.qq, t.u, kTanhQ[0]); t.e0 = __builtin_amdgcn_rcpf(1.0f + t.e0); break;      // i
            case 5: t.gg = (t.p * t.x) * __builtin_amdgcn_rcpf(t.qq); t.e1 = __builtin_amdgcn_rcpf(1.0f + t.e1); break;                                                // tanh(g), f
            case 6: creg[tl] = __builtin_fmaf(t.e1, creg[tl], t.e0 * t.gg); t.x = tanh_rat_clamp(creg[tl]); break;
            case 7: t.u = t.x * t.x; t.p = __builtin_fmaf(kTanhP[6], t.u, kTanhP[5]); t.p = __builtin_fmaf(t.p, t.u, kTanhP[4]); t.e2 = __builtin_amdgcn_rcpf(1.0f + t.e2); break;   // o
            case 8: t.p = __builtin_fmaf(t.p, t.u, kTanhP[3]); t.p = __builtin_fmaf(t.p, t.u, kTanhP[2]); t.p = __builtin_fmaf(t.p, t.u, kTanhP[1]); break;
            case 9: t.p = __builtin_fmaf(t.p, t.u, kTanhP[0]); t.qq = __builtin_fmaf(kTanhQ[3], t.u, kTanhQ[2]); t.qq = __builtin_fmaf(t.qq, t.u, kTanhQ[1]); t.qq = __builtin_fmaf(t.qq, t.u, kTanhQ[0]); break;
            case 10: hv[tl] = t.e2 * ((t.p * t.x) * __builtin_amdgcn_rcpf(t.qq)); break;
            case 11: publish_tile(h_dst, tl); break;
            default: break;
            }
        } else {                                            // GRU: g = (z, r, candidate's recurrent half, its input half)
            switch (k) {
            case 0: t.e0 = __builtin_amdgcn_exp2f(g.x); t.e1 = __builtin_amdgcn_exp2f(g.y); break;
            case 1: t.e0 = __builtin_amdgcn_rcpf(1.0f + t.e0); t.e1 = __builtin_amdgcn_rcpf(1.0f + t.e1); break;
            case 2: t.e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.e1, g.z, g.w)); break;                  // (the candidate rows carry 2 log2 e: tanh_exp_pre)
            case 3: t.x = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + t.e2), 1.0f); hv[tl] = __builtin_fmaf(t.e0, hv[tl] - t.x, t.x); break;
            case 4: publish_tile(h_dst, tl); break;
            default: break;
            }
        }
    };

    // A frame goes up the ring in two parts: its h fragments one tick after it was computed (16-byte write-through stores out
    // of LDS), the tiles started for the layer above another tick later — they are computed at the END of a tick (the MFMAs
    // whose shadow the cell update issues in) and leave at the head of the next one, so that every store has a whole tick to
    // drain before the barrier that precedes its publication.
    auto ship_h = [&](const u32x4* h_src, int Fprev) {
#ifndef AIDAX_LS_NOSHIP                                              // (measurement build without the ring's stores: what the hand-over costs the lower layer)
        const uint32_t slot_off = (uint32_t)(((base_out + (uint32_t)Fprev) % kLpRing) * kSlot * sizeof(float));
        for (int i = tid; i < kFrag; i += NT) lp_store16(rs_out, slot_off + (uint32_t)i * 16u, __builtin_bit_cast(f32x4, h_src[i]));
#else
        (void)h_src; (void)Fprev;
#endif
    };
    auto ship_started = [&](const f32x4 (&pacc)[MA], int Fprev) {
#ifndef AIDAX_LS_NOSHIP
        const uint32_t slot_off = (uint32_t)(((base_out + (uint32_t)Fprev) % kLpRing) * kSlot * sizeof(float));
#pragma unroll
        for (int tl = 0; tl < MW; ++tl)
            lp_store16(rs_out, slot_off + (uint32_t)(kFrag * 16) + (uint32_t)(((wave * M + tl) * kWave + lane) * 16), pacc[tl]);
#else
        (void)pacc; (void)Fprev;
#endif
    };
    auto no_valu = [](int) {};
    auto own_w = [&](int tl, int ks, int tw) -> bf16x8 { return wq[tl][ks][tw]; };
    // the segments nobody waits for: x = tile (first layer: the tiles started for the layer above) / tile - MW (others)
#ifdef AIDAX_LS_NOXW                                                 // (measurement build, wrong results: what reading the LDS-resident terms at use costs)
    auto x_w = [&](int x, int ks, int tw) -> bf16x8 { (void)tw; return wx[x][ks]; };
#else
    auto x_w = [&](int x, int ks, int tw) -> bf16x8 { return tw == 0 ? wx[x][ks] : __builtin_bit_cast(bf16x8, w2x[((x * KS2 + ks) * 2 + tw - 1) * 64]); };
#endif
    auto up_w = [&](int tl, int ks, int tw) -> bf16x8 { return x_w(tl, ks, tw); };
    auto below_w = [&](int tl, int ks, int tw) -> bf16x8 { return x_w(tl - MW, ks, tw); };
    auto bias_of = [&](int tl, bool above) { return *reinterpret_cast<const f32x4*>(biasL + (above ? H * 4 : 0) + 4 * (4 * (wave * TPW + tl) + q)); };

    constexpr int HALF = (TPW + 1) / 2;                    // tiles of the first own-h phase
    constexpr int G = ls_gate_groups(KS2, NPROD);          // MFMA groups of one ls_gates call
    f32x4 held[MA] = {};                                   // first layer: the started tiles of the frame before last, on their way out
    int par = 0;                                           // parity of hT the next frame reads
    int done = 0;                                          // frames finished before this chunk
    for (int base = 0; base < n; base += kLpChunk) {
        const int cnt = n - base < kLpChunk ? n - base : kLpChunk;
        auto wait_below = [&](int frames_needed) {        // thread 0 only: until `frames_needed` frames of the launch exist
            if (frames_needed > n) frames_needed = n;
            uint32_t spins = 0;
            uint64_t t0 = 0;
            while ((int)known_below < frames_needed) {
                known_below = __hip_atomic_load(cnt_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base_in;
                if ((int)known_below < frames_needed) {
                    if (lp_timed_out(spins, t0)) { give_up(); known_below = (uint32_t)n; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        };
        // ---- audio rows of this chunk: the first layer reads them as input, the last for in_skip and to deliver
        if constexpr (chain && last && !first) {
            // one-launch form: the rows in a.out are the pre pass's, stored (write-through, drained) by the FIRST layer's
            // workgroup before it computed its first frame — once frames exist below, the rows are there
            if (tid == 0) wait_below(done + 3);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (first || last) {
            for (int sl = wave; sl < NS; sl += NW) {
                const int sg = s_base + sl;
                const bool lv = livef[sl] != 0.f;
                float* row = xb + sl * nP;
                if (lv) {
                    const float* src = a.out + (size_t)sg * n + base;
                    if (((n | base) & 3) == 0) load_block(row, src, cnt, lane);
                    else for (int t = lane; t < cnt; t += kWave) row[t] = src[t];
                } else {
                    for (int t = lane; t < cnt; t += kWave) row[t] = 0.f;
                }
            }
        }
        __syncthreads();
        auto write_xin = [&](int parity, int f) {
            // lanes tid < NS: x * in_gain, PARAM1, PARAM2 of frame `f` (:171-181, :195-231)
            float q1 = 0.f, q2 = 0.f;
            if (I >= 2) q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
            if (I >= 3) q2 = lin_next(p_mem[1], p_tgt[1], p_step[1]);
            float* col = xin + parity * 64;
            col[tid] = xb[tid * nP + f] * a.in_gain;
            col[NS + tid] = q1;
            col[2 * NS + tid] = q2;
            col[3 * NS + tid] = 0.f;
        };
        constexpr int PER4 = (kFrag + NT - 1) / NT;        // 16-byte vectors of a frame each thread moves
        f32x4 pre[PER4];
        f32x4 upx[MA] = {};                                // started tiles on their way in (layers >= 1): fetched in one tick, the accumulators' start in the next
        constexpr int NB = TPW - MW > 0 ? TPW - MW : 1;
        f32x4 nacc[NB] = {};                               // layers >= 1: bias + input half of the tiles that start here, for the NEXT frame
        auto fetch_frags = [&](int F) {                    // all threads: the h fragments of frame F of the launch -> registers
            const uint32_t slot_off = (uint32_t)(((base_in + (uint32_t)F) % kLpRing) * kSlot * sizeof(float));
#pragma unroll
            for (int k = 0; k < PER4; ++k)
                if (k * NT + tid < kFrag) pre[k] = lp_load16(rs_in, slot_off + (uint32_t)(k * NT + tid) * 16u);
        };
        auto fetch_started = [&](int F) {                  // ... and this wave's started tiles of frame F
            const uint32_t slot_off = (uint32_t)(((base_in + (uint32_t)F) % kLpRing) * kSlot * sizeof(float));
#pragma unroll
            for (int tl = 0; tl < MW; ++tl)
                upx[tl] = lp_load16(rs_in, slot_off + (uint32_t)(kFrag * 16) + (uint32_t)(((wave * M + tl) * kWave + lane) * 16));
        };
        auto stash_below = [&](int parity) {
#pragma unroll
            for (int k = 0; k < PER4; ++k)
                if (k * NT + tid < kFrag) below[parity * kFrag + k * NT + tid] = __builtin_bit_cast(u32x4, pre[k]);
        };
        // the input half of the tiles that start here, for the frame whose fragments sit in below[parity]
        auto start_next = [&](int parity, auto valu) {
            f32x4 z[TPW];
#pragma unroll
            for (int tl = MW; tl < TPW; ++tl) z[tl] = bias_of(tl, false);
            LsFrags<KS2> hbn;
            ls_load_frags(hbn, below + parity * kFrag, lane, AIDAX_TUNE(a));
            ls_gates<KS2, NPROD, MW, TPW, TPW>(z, hbn, below_w, valu);
#pragma unroll
            for (int tl = MW; tl < TPW; ++tl) nacc[tl - MW] = z[tl];
        };

        if (first) {
            if (tid < NS) write_xin(0, 0);
        } else {
            // frames 0 and 1 of the chunk into both halves of `below`, frame 0's started tiles, frame 2 for tick 0's prefetch
            if (tid == 0) wait_below(done + 3);
            __syncthreads();
            fetch_frags(done);
            fetch_started(done);
            stash_below(0);
            if (cnt > 1) { fetch_frags(done + 1); stash_below(1); }
        }
        __syncthreads();
        if constexpr (!first && MW < TPW) start_next(0, no_valu);

        const int ticks = last ? cnt + 2 : cnt;            // the Dense of a frame: partial sums one tick behind its h, the output two
        for (int tick = 0; tick < ticks; ++tick) {
            const int rd = par, wr = par ^ 1;
            const bool body = tick < cnt;
            const bool more = tick + 1 < cnt;              // another frame of this chunk follows
            const bool more2 = tick + 2 < cnt;
            const int F = done + tick;                     // frame of the launch this tick computes
            LS_STAMP(0);
            f32x4 acc[TPW];
            if (first) {
                if (tid < NS && more) write_xin((tick + 1) & 1, tick + 1);
            } else {
                // bias + input half: the tiles that arrived started (fetched a tick ago) / the ones started here last tick
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) acc[tl] = tl < MW ? upx[tl < MW ? tl : 0] : nacc[tl < MW ? 0 : tl - MW];
                if (more) fetch_started(F + 1);
                if (more2) fetch_frags(F + 2);
            }
            if (body && tid == 0) {
                // thread 0 looks ahead while the others compute (see k_mfma_lp): what the NEXT tick fetches must exist below,
                // the slot the next tick stores into must be free above
                if (!first && tick + 3 < cnt) wait_below(F + 4);
                if (!last && base + tick + 1 < n) {
                    uint32_t spins = 0;
                    uint64_t t0 = 0;
                    while ((int)known_free < F + 2) {
                        const uint32_t consumed = __hip_atomic_load(cnt_out + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base_out;
                        known_free = consumed + kLpRing;
                        if ((int)known_free < F + 2) {
                            if (lp_timed_out(spins, t0)) { give_up(); known_free = (uint32_t)n + kLpRing; break; }
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                }
            }
            // (a lone layer: the tick's fragment reads are requested before the Dense, whose 340 cycles then pass while they arrive —
            // 48 KiB for the eight waves of LSTM-64, 384 cycles of the LDS port; the stacked roles have no registers to spare for that)
            constexpr bool early_frags = first && last;
            LsFrags<KS2> hb;                                // own h(t-1)
            if constexpr (early_frags) { if (body) ls_load_frags(hb, hT + rd * kFrag, lane, AIDAX_TUNE(a)); }
            // ---- Dense(H,1) of frame tick-1 from the lanes' own h(tick-1): this wave's units, summed in a fixed order
            if (last && tick >= 1 && tick <= cnt) {
                float y = wdu[0] * hv[0];
#pragma unroll
                for (int tl = 1; tl < TPW; ++tl) y = __builtin_fmaf(wdu[tl], hv[tl], y);
                const Pair r2 = share_rows(y);
                y = r2.lo + r2.hi;
                const Pair r4 = share_halves(y);
                y = r4.lo + r4.hi;
                if (lane < NS) dpart[((tick & 1) * NW + wave) * NS + lane] = y;
            }
            // ---- ... and of frame tick-2: the NW partial sums, bias, skip, output gain (wave 0, one lane per stream)
            if (last && tick >= 2 && wave == (early_frags && NW > 1 ? 1 : 0) && lane < NS) {      // (lone layer: not on the wave that writes the model inputs)
                const int fd = tick - 2;
                float y = wdl[H];
#pragma unroll
                for (int w = 0; w < NW; ++w) y += dpart[(((tick - 1) & 1) * NW + w) * NS + lane];
                const float x = xb[lane * nP + fd] * a.in_gain;
                float o = a.input_skip ? x + y : y;
                o = o * a.out_gain;
                if (livef[lane] != 0.f) xb[lane * nP + fd] = o;
            }

            LS_STAMP(1);
            if (body) {
                const u32x4* h_rd = hT + rd * kFrag;
                u32x4* h_wr = hT + wr * kFrag;
                if constexpr (!early_frags) ls_load_frags(hb, h_rd, lane, AIDAX_TUNE(a));      // requested first, on its way while the frame before goes up the ring
                if (!last && F >= 1) ship_h(h_rd, F - 1);              // h_rd = h(F-1)
                if constexpr (first) {
                    if (MW > 0 && F >= 2) ship_started(held, F - 2);
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) acc[tl] = bias_of(tl, false);
                    const float b = xin[(tick & 1) * 64 + lane];      // the model inputs: one fp32 k-step (x, PARAM1, PARAM2, 0)
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl)
                        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in0[tl], b, acc[tl], 0, 0, 0);
                }
                // the cell steps of the tiles [T0, T1), dealt out over the G groups of a phase (the tiles advance together)
                auto deal = [&](int T0, int T1, int g) {
                    const int nt = T1 - T0, S = nt * NSTEP;
#pragma unroll
                    for (int sidx = g * S / G; sidx < (g + 1) * S / G; ++sidx) {
#ifdef AIDAX_LS_SEQ_CELLS                                           // (one tile's update after the other: eight temporaries fewer, less to overlap)
                        cell_step(T0 + sidx / NSTEP, sidx % NSTEP, acc[T0 + sidx / NSTEP], h_wr);
#else
                        cell_step(T0 + sidx % nt, sidx / nt, acc[T0 + sidx % nt], h_wr);
#endif
                    }
                };
                // phase A: own h(t-1) against the first half of the tiles
                LS_STAMP(2);
                ls_gates<KS2, NPROD, 0, HALF, TPW>(acc, hb, own_w, no_valu);
                LS_STAMP(3);
#ifndef AIDAX_LS_STASH_LATE
                if constexpr (!first) {
                    // frame F+2's fragments (requested at the head of the tick) go where frame F's were: their product was taken
                    // a tick ago, and the registers they travelled in are free for the phases below. (At the END of the tick —
                    // AIDAX_LS_STASH_LATE — the load has the whole tick to arrive: measured the same for LSTM-96 x 2, 712.5
                    // against 713.3 us, and 3 % slower for LSTM-64 x 2.)
                    if (more2) stash_below(tick & 1);
                }
#endif
                // phase B: ... the second half, the first half's cell update in its shadow
                if constexpr (HALF < TPW) ls_gates<KS2, NPROD, HALF, TPW, TPW>(acc, hb, own_w, [&](int g) { deal(0, HALF, g); });
                else {
#pragma unroll
                    for (int g = 0; g < G; ++g) deal(0, HALF, g);
                }
                LS_STAMP(4);
                // phase C: MFMAs nobody waits for — the tiles started for the layer above (first layer) / the input half of the
                // next frame (others) — with the second half's cell update in their shadow
                bool c_done = false;
                if constexpr (first) {
                    if constexpr (MW > 0) {
                        if (F >= 1) {
                            f32x4 pacc[MA];
#pragma unroll
                            for (int tl = 0; tl < MW; ++tl) pacc[tl] = bias_of(tl, true);
                            ls_gates<KS2, NPROD, 0, MW, MA>(pacc, hb, up_w, [&](int g) { if constexpr (HALF < TPW) deal(HALF, TPW, g); });
#pragma unroll
                            for (int tl = 0; tl < MA; ++tl) held[tl] = pacc[tl];
                            c_done = true;
                        }
                    }
                } else if constexpr (MW < TPW) {
                    if (more) {
                        start_next((tick + 1) & 1, [&](int g) { if constexpr (HALF < TPW) deal(HALF, TPW, g); });
                        c_done = true;
                    }
                }
                if (!c_done) {
                    if constexpr (HALF < TPW) {
#pragma unroll
                        for (int g = 0; g < G; ++g) deal(HALF, TPW, g);
                    }
                }
                LS_STAMP(5);
#ifdef AIDAX_LS_STASH_LATE
                if constexpr (!first) { if (more2) stash_below(tick & 1); }
#endif
                par = wr;
                if (!last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's ring stores (issued at the head of the tick) have left before the barrier (G16)
                LS_STAMP(6);
            }
            __syncthreads();                               // h(t) in LDS, the next frames' input in LDS, ring stores of this tick done
            LS_STAMP(7);
            // ---- counters, by thread 0 after the barrier: every kLpBatch frames and at the end of the launch
            if (body && tid == 0) {
                if (!last) {
                    // complete in the ring: frames whose h left by the last tick and whose started tiles left in this one
                    const int produced = first && MW > 0 ? F - 1 : F;
                    if (produced > 0 && produced % kLpBatch == 0)
                        __hip_atomic_store(cnt_out, base_out + (uint32_t)produced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!first) {
                    const int consumed = more ? F + 2 : F + 1;             // frames read out of the ring so far (fragments AND started tiles)
                    if (consumed % kLpBatch == 0 || consumed == n)
                        __hip_atomic_store(cnt_in + 16, base_in + (uint32_t)consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        done += cnt;
        // ---- results of this chunk back to HBM (last layer)
        if (last) {
            for (int sl = wave; sl < NS; sl += NW) {
                const int sg = s_base + sl;
                if (livef[sl] == 0.f) continue;
                float* dst = a.out + (size_t)sg * n + base;
                const float* row = xb + sl * nP;
                if (((n | base) & 3) == 0) store_block(dst, row, cnt, lane);
                else for (int t = lane; t < cnt; t += kWave) dst[t] = row[t];
            }
        }
        __syncthreads();
    }

    if (!last && n > 0) {                                  // the launch's last frame (and the started tiles still held), then the final count
        ship_h(hT + par * kFrag, n - 1);
        if constexpr (first && MW > 0) {
            if (n >= 2) ship_started(held, n - 2);
            f32x4 pacc[MA];
#pragma unroll
            for (int tl = 0; tl < MW; ++tl) pacc[tl] = bias_of(tl, true);
            LsFrags<KS2> hbl;
            ls_load_frags(hbl, hT + par * kFrag, lane);
            ls_gates<KS2, NPROD, 0, MW, MA>(pacc, hbl, up_w, no_valu);
            ship_started(pacc, n - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(cnt_out, base_out + (uint32_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- recurrent state and smoother memories back to HBM for the streams that ran
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        hS[(wave * TPW + tl) * 64 + lane] = hv[tl];
        cT[(wave * TPW + tl) * 64 + lane] = creg[tl];
    }
    __syncthreads();
    for (int i = tid; i < H * NS; i += NT) {
        const int u = i / NS, sn = i % NS, sg = s_base + sn;
        if (sg < (int)a.n_streams && u < Ht && livef[sn] != 0.f) {
            float* stp = a.nn + (size_t)sg * a.nn_stride + L.state_off;
            stp[u] = hS[i];
            if (L.cell == 0) stp[Ht + u] = cT[i];
        }
    }
    if (first && tid < NS && mine_live) {
        StreamState& st = a.st[s_base + tid];
        st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
        st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
        st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
        if (!chain) st.pending = pending;
        else if (pending != st_pending0)                  // (the last layer's workgroup clears PEND_ACTIVATE in the same word)
            __hip_atomic_fetch_and(&st.pending, ~(uint32_t)PEND_PARAM_FIRST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int TPW, int NW, int M, bool CHAIN, int NPROD>
__global__ __launch_bounds__(NW * kWave) void k_mfma_ls(LaunchArgs a, MfmaDesc d, float* ring, uint32_t* counters, uint32_t* fault)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NL = d.n_layers;
    const int blk = (int)blockIdx.x;
    const bool adjacent = (AIDAX_TUNE(a) & 2) != 0;
    const int grp = adjacent ? blk / NL : (blk / (8 * NL)) * 8 + (blk & 7);
    const int l = adjacent ? blk % NL : (blk / 8) % NL;
    const int n_groups = ((int)a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    if (grp >= n_groups) return;
    // (measurement, AIDAX_TUNE bit 4096: where this workgroup ran and when — XCC_ID, HW_ID, the 100 MHz clock at its start and end — into
    // the pool's pinned fault page behind the fault word; scratch/r05_modes.py reads it)
    const bool stamp = (AIDAX_TUNE(a) & 4096) && blk < 500 && threadIdx.x == 0;
    if (stamp) {
        fault[16 + 3 * blk] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) << 4);
        fault[17 + 3 * blk] = (uint32_t)wall_clock64();
    }
    if (l == 0) {
        if constexpr (CHAIN) {
            lp_chain_rows<true>(a, smem, grp, true);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (d.L[l].cell == 0) ls_body<TPW, NW, M, true, false, CHAIN, NPROD, 0>(a, d, ring, counters, fault, smem, grp, l);
        else ls_body<TPW, NW, M, true, false, CHAIN, NPROD, 1>(a, d, ring, counters, fault, smem, grp, l);
    } else if (l == NL - 1) {
        if constexpr (CHAIN) {
            if (AIDAX_TUNE(a) & 8192) {                                          // test hook: this workgroup starts 100 us late
                const uint64_t t0 = wall_clock64();
                while (wall_clock64() - t0 < 10000) __builtin_amdgcn_s_sleep(64);
            }
        }
        if (d.L[l].cell == 0) ls_body<TPW, NW, M, false, true, CHAIN, NPROD, 0>(a, d, ring, counters, fault, smem, grp, l);
        else ls_body<TPW, NW, M, false, true, CHAIN, NPROD, 1>(a, d, ring, counters, fault, smem, grp, l);
        if constexpr (CHAIN) {
            __syncthreads();
            lp_chain_rows<false>(a, smem, grp, true);
        }
    } else {
        if (d.L[l].cell == 0) ls_body<TPW, NW, M, false, false, CHAIN, NPROD, 0>(a, d, ring, counters, fault, smem, grp, l);
        else ls_body<TPW, NW, M, false, false, CHAIN, NPROD, 1>(a, d, ring, counters, fault, smem, grp, l);
    }
    if (stamp) fault[18 + 3 * blk] = (uint32_t)wall_clock64();
}

// A LONE layer on the same body (first and last role at once: no ring, nobody waits, a workgroup per 16 streams): the one-layer
// models of the reference's table at stream counts where a matrix-core form pays — LSTM-64 / 80, GRU-80, ... (GRU-40 / 64 have
// k_gru_gs). CHAIN: the DSP chain passes on waves 0 and 1 around the body, the whole run() in the launch.
template <int TPW, int NW, bool CHAIN, int NPROD>
__global__ __launch_bounds__(NW * kWave) void k_mfma_ls1(LaunchArgs a, MfmaDesc d, float* ring, uint32_t* counters, uint32_t* fault)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = (int)blockIdx.x;
    if constexpr (CHAIN) {
        lp_chain_rows<true>(a, smem, grp, true);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (d.L[0].cell == 0) ls_body<TPW, NW, 0, true, true, CHAIN, NPROD, 0>(a, d, ring, counters, fault, smem, grp, 0);
    else ls_body<TPW, NW, 0, true, true, CHAIN, NPROD, 1>(a, d, ring, counters, fault, smem, grp, 0);
    if constexpr (CHAIN) {
        __syncthreads();
        lp_chain_rows<false>(a, smem, grp, true);
    }
}

// ---------------------------------------------------------------- host side
// The file compiles as three objects (Makefile: -DAIDAX_MFMALP_PART=1 .. 4), each instantiating its own share of the kernel
// templates above — 1: k_mfma_lp, k_gru_gm; 2 and 4: k_mfma_ls, k_mfma_ls1; 3: k_gru_gs, k_lstm_gs — so that a clean build does not wait
// four minutes for one translation unit (part 4: the two-layer instantiations of k_mfma_ls, the largest group). Without the macro everything lands in one object (the measurement builds of scratch/).
#ifndef AIDAX_MFMALP_PART
#define AIDAX_MFMALP_PART 0
#endif
#define AIDAX_PART(n) (AIDAX_MFMALP_PART == 0 || AIDAX_MFMALP_PART == (n))
typedef void (*LpFn)(LaunchArgs, MfmaDesc, float*, uint32_t*, uint32_t*);
typedef void (*GmFn)(LaunchArgs, MfmaDesc);
[[maybe_unused]] constexpr int kLpHelpers = 4;
// The chained kernels (two or more layers: workgroups that wait for each other) go out through hipLaunchCooperativeKernel: the
// runtime REFUSES a grid that cannot be co-resident on the device (hipErrorCooperativeLaunchTooLarge — the pool then serves the
// model with k_mfma at once, aidax_pool.cpp) and schedules the grid as a gang, so that another process's kernels cannot sit
// between its workgroups: next to a process that keeps every CU busy a cfg5 block took 7.8 ms through the plain launch and
// 0.71 ms through this one; alone it costs 15 us per block (698 -> 713 us, profiles/r04_lp_coop.txt). AIDAX_LP_COOP=0: the
// plain launch (A/B runs).
[[maybe_unused]] static bool lp_coop_launch()
{
    static const bool on = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_LP_COOP"); return !(e && e[0] == '0'); }();
    return on;
}
[[maybe_unused]] static hipError_t lp_launch(LpFn fn, uint32_t blocks, uint32_t threads, size_t lds, hipStream_t stream, LaunchArgs a, MfmaDesc d,
                            float* ring, uint32_t* counters, uint32_t* fault)
{
    if (lp_coop_launch() && d.n_layers >= 2) {
        void* args[] = { &a, &d, &ring, &counters, &fault };
        return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(blocks), dim3(threads), args, (unsigned)lds, stream);
    }
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, stream, a, d, ring, counters, fault);
    return hipGetLastError();
}
#if AIDAX_PART(1)
// Helper waves of the one-launch form: a third wave per SIMD must fit next to the main waves' registers (512 per SIMD
// lane: H <= 64 at <= 160 registers; H = 80 / 96 hold 256 / 233 and keep the three-launch form).
static int lp_helpers(int hidden) { return hidden == 16 || hidden == 32 || hidden == 48 || hidden == 64 ? kLpHelpers : 0; }
static LpFn lp_fn_fused(int hidden)
{
    switch (hidden) {
#define AIDAX_LP_FUSED_CASE(HID) case HID: { constexpr int T = HID / 4 / mfma_waves(HID), W_ = mfma_waves(HID); return k_mfma_lp<T, W_, 0, kLpHelpers>; }
    AIDAX_LP_FUSED_CASE(16) AIDAX_LP_FUSED_CASE(32) AIDAX_LP_FUSED_CASE(48) AIDAX_LP_FUSED_CASE(64)
#undef AIDAX_LP_FUSED_CASE
    default: return nullptr;
    }
}
static LpFn lp_fn_chain(int hidden, int n_layers)         // the DSP chain on waves 0 and 1 of the first / last layer's workgroup
{
    switch (hidden) {
#define AIDAX_LP_CHAIN_CASE(HID) case HID: { constexpr int T = HID / 4 / mfma_waves(HID), W_ = mfma_waves(HID);                    \
        return lp_moved_tiles(2, T, W_) > 0 && n_layers == 2 ? k_mfma_lp<T, W_, lp_moved_tiles(2, T, W_), 0, true> : k_mfma_lp<T, W_, 0, 0, true>; }
    AIDAX_LP_CHAIN_CASE(16) AIDAX_LP_CHAIN_CASE(32) AIDAX_LP_CHAIN_CASE(48) AIDAX_LP_CHAIN_CASE(64) AIDAX_LP_CHAIN_CASE(80) AIDAX_LP_CHAIN_CASE(96)
#undef AIDAX_LP_CHAIN_CASE
    default: return nullptr;
    }
}
static LpFn lp_fn(int hidden, int n_layers)
{
    switch (hidden) {
#define AIDAX_LP_CASE(HID) case HID: { constexpr int T = HID / 4 / mfma_waves(HID), W_ = mfma_waves(HID);                         \
        return lp_moved_tiles(2, T, W_) > 0 && n_layers == 2 ? k_mfma_lp<T, W_, lp_moved_tiles(2, T, W_)> : k_mfma_lp<T, W_, 0>; }
    AIDAX_LP_CASE(16) AIDAX_LP_CASE(32) AIDAX_LP_CASE(48) AIDAX_LP_CASE(64) AIDAX_LP_CASE(80) AIDAX_LP_CASE(96)
#undef AIDAX_LP_CASE
    default: return nullptr;                               // wider stacks keep the fragment-streaming kernel
    }
}
static int lp_m(const MfmaDesc& d) { return lp_moved_tiles(d.n_layers, d.hidden / 4 / mfma_waves(d.hidden), mfma_waves(d.hidden)); }

// (one-layer models too: a workgroup per 16 streams with the layer's fragments resident in registers — no ring, no waits)
bool mfma_lp_serves(const MfmaDesc& d) { return d.n_layers >= 1 && lp_fn(d.hidden, d.n_layers) != nullptr; }
// which one-launch form a model takes: helper waves (one layer, room for a third wave per SIMD) or the chain passes on two
// main waves around the body (everything else)
static bool lp_helper_form(const MfmaDesc& d) { return d.n_layers == 1 && lp_helpers(d.hidden) > 0; }
size_t mfma_lp_lds_bytes(const MfmaDesc& d, uint32_t n_frames, bool fused)
{
    if (fused && !lp_helper_form(d)) {                     // lp_chain_rows works in the body's LDS before and after it
        const size_t body = lp_lds_floats(d.hidden, (int)n_frames), rows = (size_t)kMfmaStreams * ((n_frames + 3) & ~3u) + 2 * kChainPackHandFloats;
        return (body > rows ? body : rows) * sizeof(float);
    }
    return lp_lds_floats(d.hidden, (int)n_frames, fused ? lp_helpers(d.hidden) : 0) * sizeof(float);
}
// the whole run() in the one launch (a.in -> a.out, MODE_CHAIN): one-layer models with room for the helper waves; all
// others (stacked, or one layer of 80 / 96 units) when their blocks fit one staging chunk (lp_chain_rows around the body)
bool mfma_lp_fused_serves(const MfmaDesc& d, uint32_t max_frames)
{
    return lp_helper_form(d) || (max_frames <= (uint32_t)kLpChunk && lp_fn_chain(d.hidden, d.n_layers) != nullptr);
}
size_t mfma_lp_ring_bytes(const MfmaDesc& d, uint32_t n_streams)
{
    const size_t groups = (n_streams + kMfmaStreams - 1) / kMfmaStreams;
    return 256 + groups * (size_t)(d.n_layers - 1) * lp_ring_floats(d.hidden, mfma_waves(d.hidden), lp_m(d)) * sizeof(float);
}
size_t mfma_lp_counter_bytes(const MfmaDesc& d, uint32_t n_streams)
{
    const size_t groups = (n_streams + kMfmaStreams - 1) / kMfmaStreams;
    return 256 + groups * (size_t)(d.n_layers - 1) * kLpCounterStride * sizeof(uint32_t);
}

hipError_t launch_mfma_lp_kernel(const LaunchArgs& a, const MfmaDesc& d, float* ring, uint32_t* counters, uint32_t* fault, hipStream_t stream, bool fused)
{
    if (fused && (!mfma_lp_fused_serves(d, a.n_frames) || a.mode != MODE_CHAIN || a.n_frames == 0)) return hipErrorInvalidValue;
    LpFn fn = !fused ? lp_fn(d.hidden, d.n_layers) : lp_helper_form(d) ? lp_fn_fused(d.hidden) : lp_fn_chain(d.hidden, d.n_layers);
    if (!fn || !ring || !counters || !fault) return hipErrorInvalidValue;
    const size_t lds = mfma_lp_lds_bytes(d, a.n_frames, fused);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    const uint32_t blocks = ((groups + 7) / 8) * 8 * (uint32_t)d.n_layers;
    const int waves = mfma_waves(d.hidden) + (fused && lp_helper_form(d) ? lp_helpers(d.hidden) : 0);
    return lp_launch(fn, blocks, (uint32_t)(waves * kWave), lds, stream, a, d, ring, counters, fault);
}

// k_gru_gm: one-layer GRU models with three or four main waves (48 / 64 units after rounding up to 16). Narrower ones would
// leave SIMDs without a main wave — GRU-32 at 4096 streams: 245 us here against 197 us on k_mfma_lp's one-launch form,
// whose eight waves share one tile row each.
static GmFn gm_fn(int hidden)
{
    switch (hidden) {
    case 48: return k_gru_gm<3, kLpHelpers>;
    case 64: return k_gru_gm<4, kLpHelpers>;
    default: return nullptr;
    }
}
bool gru_gm_serves(const MfmaDesc& d) { return d.n_layers == 1 && d.L[0].cell == 1 && d.gm_off != 0 && gm_fn(d.hidden) != nullptr; }
size_t gru_gm_lds_bytes(const MfmaDesc& d, uint32_t n_frames) { return gm_lds_floats(d.hidden, (int)n_frames, kLpHelpers) * sizeof(float); }
hipError_t launch_gru_gm_kernel(const LaunchArgs& a, const MfmaDesc& d, hipStream_t stream)
{
    GmFn fn = gm_fn(d.hidden);
    if (!fn || !gru_gm_serves(d) || a.mode != MODE_CHAIN || a.n_frames == 0) return hipErrorInvalidValue;
    const size_t lds = gru_gm_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    hipLaunchKernelGGL(fn, dim3(groups), dim3((d.hidden / 16 + kLpHelpers) * kWave), lds, stream, a, d);
    return hipGetLastError();
}

#endif      // part 1
#if AIDAX_PART(3)
// k_gru_gs: the same models as k_gru_gm, recurrent product on the bf16 matrix pipe (operands split into three bf16 terms)
static GmFn gs_fn(int hidden, int nprod)
{
    switch (hidden) {
    case 48: return nprod == 9 ? k_gru_gs<3, kLpHelpers, 9> : k_gru_gs<3, kLpHelpers, 6>;
    case 64: return nprod == 9 ? k_gru_gs<4, kLpHelpers, 9> : k_gru_gs<4, kLpHelpers, 6>;
    // 80 units: five main waves and TWO helpers (eight streams each) — seven waves, two per SIMD at most, 256 registers each;
    // one SIMD carries two main waves (no fp32 twin: k_gru_gm stays with 48 / 64 units)
    case 80: return nprod == 9 ? k_gru_gs<5, 2, 9> : k_gru_gs<5, 2, 6>;
    default: return nullptr;
    }
}
static int gs_helpers(int hidden) { return hidden == 80 ? 2 : kLpHelpers; }
bool gru_gs_serves(const MfmaDesc& d) { return d.n_layers == 1 && d.L[0].cell == 1 && d.gm_off != 0 && d.gs_off != 0 && gs_fn(d.hidden, 6) != nullptr; }
size_t gru_gs_lds_bytes(const MfmaDesc& d, uint32_t n_frames) { return gs_lds_floats(d.hidden, (int)n_frames, gs_helpers(d.hidden)) * sizeof(float); }
hipError_t launch_gru_gs_kernel(const LaunchArgs& a, const MfmaDesc& d, int n_products, hipStream_t stream)
{
    GmFn fn = gs_fn(d.hidden, n_products);
    if (!fn || !gru_gs_serves(d) || a.mode != MODE_CHAIN || a.n_frames == 0) return hipErrorInvalidValue;
    const size_t lds = gru_gs_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    hipLaunchKernelGGL(fn, dim3(groups), dim3((d.hidden / 16 + gs_helpers(d.hidden)) * kWave), lds, stream, a, d);
    return hipGetLastError();
}

// k_lstm_gs: one-layer LSTMs of 48 (LSTM-40 padded) / 64 units — as many main waves as a CU has SIMDs or one fewer, 256 registers each
static GmFn lgs_fn(int hidden, int nprod)
{
    switch (hidden) {
    case 48: return nprod == 9 ? k_lstm_gs<3, kLpHelpers, 9> : k_lstm_gs<3, kLpHelpers, 6>;
    case 64: return nprod == 9 ? k_lstm_gs<4, kLpHelpers, 9> : k_lstm_gs<4, kLpHelpers, 6>;
    case 80: return k_lstm_gs<5, 2, 6>;                      // (five main waves, two helpers: as k_gru_gs for 80 units; nine products do not fit the registers)
    default: return nullptr;
    }
}
bool lstm_gs_serves(const MfmaDesc& d) { return d.n_layers == 1 && d.L[0].cell == 0 && d.gs_off != 0 && lgs_fn(d.hidden, 6) != nullptr; }
size_t lstm_gs_lds_bytes(const MfmaDesc& d, uint32_t n_frames) { return gs_lds_floats(d.hidden, (int)n_frames, gs_helpers(d.hidden)) * sizeof(float); }
hipError_t launch_lstm_gs_kernel(const LaunchArgs& a, const MfmaDesc& d, int n_products, hipStream_t stream)
{
    GmFn fn = lgs_fn(d.hidden, n_products);
    if (!fn || !lstm_gs_serves(d) || a.mode != MODE_CHAIN || a.n_frames == 0) return hipErrorInvalidValue;
    const size_t lds = lstm_gs_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    hipLaunchKernelGGL(fn, dim3(groups), dim3((d.hidden / 16 + gs_helpers(d.hidden)) * kWave), lds, stream, a, d);
    return hipGetLastError();
}

#endif      // part 3
#if AIDAX_PART(2) || AIDAX_PART(4)
// k_mfma_ls: the stacked models k_mfma_lp serves whose resident split fragments fit the register file (ls_geo)
template <int HID, int NL>
static LpFn ls_fn_for(bool chain, int nprod)
{
    constexpr LsGeo g = ls_geo(NL, HID);
    if constexpr (g.nw == 0) return nullptr;
    else if constexpr (NL == 1) {
        if constexpr (g.tpw * ls_ks2(HID) * 12 > 150) return chain ? k_mfma_ls1<g.tpw, g.nw, true, 6> : k_mfma_ls1<g.tpw, g.nw, false, 6>;
        else return chain ? (nprod == 9 ? k_mfma_ls1<g.tpw, g.nw, true, 9> : k_mfma_ls1<g.tpw, g.nw, true, 6>)
                          : (nprod == 9 ? k_mfma_ls1<g.tpw, g.nw, false, 9> : k_mfma_ls1<g.tpw, g.nw, false, 6>);
    }
    else if constexpr ((g.tpw + g.nx) * ls_ks2(HID) * 12 > 150)      // the 512-register geometries: six products only (nine spill)
        return chain ? k_mfma_ls<g.tpw, g.nw, g.m, true, 6> : k_mfma_ls<g.tpw, g.nw, g.m, false, 6>;
    else return chain ? (nprod == 9 ? k_mfma_ls<g.tpw, g.nw, g.m, true, 9> : k_mfma_ls<g.tpw, g.nw, g.m, true, 6>)
                      : (nprod == 9 ? k_mfma_ls<g.tpw, g.nw, g.m, false, 9> : k_mfma_ls<g.tpw, g.nw, g.m, false, 6>);
}
#endif
LpFn ls_fn_two_layers(int hidden, bool chain, int nprod);      // (an object of its own: part 4)
#if AIDAX_PART(4)
LpFn ls_fn_two_layers(int hidden, bool chain, int nprod)
{
    switch (hidden) {                                       // two layers: tiles started below
#define AIDAX_LS_CASE(HID) case HID: return ls_fn_for<HID, 2>(chain, nprod);
    AIDAX_LS_CASE(16) AIDAX_LS_CASE(32) AIDAX_LS_CASE(48) AIDAX_LS_CASE(64) AIDAX_LS_CASE(80) AIDAX_LS_CASE(96)
#undef AIDAX_LS_CASE
    default: return nullptr;
    }
}
#endif
#if AIDAX_PART(2)
static LpFn ls_fn(int hidden, int n_layers, bool chain, int nprod)
{
    if (n_layers == 2) return ls_fn_two_layers(hidden, chain, nprod);
    switch (hidden) {                                       // (a lone layer; deeper stacks: no tiles started below — the same body with M = 0)
#define AIDAX_LS_CASE(HID) case HID: return n_layers == 1 ? ls_fn_for<HID, 1>(chain, nprod) : ls_fn_for<HID, 3>(chain, nprod);
    AIDAX_LS_CASE(16) AIDAX_LS_CASE(32) AIDAX_LS_CASE(48) AIDAX_LS_CASE(64) AIDAX_LS_CASE(80) AIDAX_LS_CASE(96)
#undef AIDAX_LS_CASE
    default: return nullptr;
    }
}
bool mfma_ls_serves(const MfmaDesc& d)
{
    return d.n_layers >= 1 && d.ls_off != 0 && ls_geo(d.n_layers, d.hidden).nw != 0 && ls_fn(d.hidden, d.n_layers, false, 6) != nullptr;
}
size_t mfma_ls_lds_bytes(const MfmaDesc& d, uint32_t n_frames, bool fused)
{
    const size_t body = ls_lds_floats(d.hidden, (int)n_frames, ls_geo(d.n_layers, d.hidden));
    const size_t rows = fused ? (size_t)kMfmaStreams * ((n_frames + 3) & ~3u) + 2 * kChainPackHandFloats : 0;      // lp_chain_rows works in the body's LDS
    return (body > rows ? body : rows) * sizeof(float);
}
bool mfma_ls_fused_serves(const MfmaDesc& d, uint32_t max_frames) { return mfma_ls_serves(d) && max_frames <= (uint32_t)kLpChunk; }
size_t mfma_ls_ring_bytes(const MfmaDesc& d, uint32_t n_streams)
{
    const size_t groups = (n_streams + kMfmaStreams - 1) / kMfmaStreams;
    const LsGeo g = ls_geo(d.n_layers, d.hidden);
    return 256 + groups * (size_t)(d.n_layers - 1) * ls_ring_floats(d.hidden, g.nw, g.m) * sizeof(float);
}
hipError_t launch_mfma_ls_kernel(const LaunchArgs& a, const MfmaDesc& d, float* ring, uint32_t* counters, uint32_t* fault, int n_products,
                                 hipStream_t stream, bool fused)
{
    if (!mfma_ls_serves(d) || a.mode != MODE_CHAIN || (fused && (a.n_frames == 0 || a.n_frames > (uint32_t)kLpChunk))) return hipErrorInvalidValue;
    LpFn fn = ls_fn(d.hidden, d.n_layers, fused, n_products);
    if (!fn || !ring || !counters || !fault) return hipErrorInvalidValue;
    const size_t lds = mfma_ls_lds_bytes(d, a.n_frames, fused);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    const uint32_t blocks = d.n_layers == 1 ? groups : ((groups + 7) / 8) * 8 * (uint32_t)d.n_layers;
    return lp_launch(fn, blocks, (uint32_t)(ls_geo(d.n_layers, d.hidden).nw * kWave), lds, stream, a, d, ring, counters, fault);
}
#endif      // part 2

}  // namespace aidax
