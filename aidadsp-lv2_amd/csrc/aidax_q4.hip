// aidax_q4.hip — k_lstm_q4: the whole run() of FOUR streams per workgroup, the recurrent cell on
// v_mfma_f32_4x4x1_16b_f32 and spread over the CU's four SIMDs. The latency form for pools whose streams number about
// four per CU (BASELINE cfg2: 1024 streams on 256 CUs), where every frame is a serial chain and the only thing that
// shortens it is fewer instructions per wave (profiles/r03_cfg2_split_ab.txt).
//
// Why this shape. At four streams per CU the one-wave-per-stream cell (k_lstm_pipe) spends 66 of its 108 instructions
// per frame on FMAs that a lone wave issues at one per ~5.5 cycles. Splitting the cell over two waves halves only the
// FMAs and pays a per-frame hand-over that costs more (measured: 16 % slower). Making the FMAs dense does pay: with the
// CU's four streams as the four columns of the 4x4x1 instruction, a wave that owns 8 hidden units x 4 gate rows and
// splits the contraction in two halves over its 16 blocks needs 17 MFMAs (148 cycles of matrix pipe) for what were 66
// FMAs of four waves; four such waves — one per SIMD — cover LSTM-32:
//   block b of wave w:  unit 8w + (b & 7), K-half b >> 3;  rows = the unit's gates i, f, g, o;  columns = streams
//   A  lane (b, row):   the gate row's weight for the half's m-th column, ONE REGISTER PER k-STEP (17), resident
//   B  lane (b, j):     x_j (step 0) resp. h_j[16 (b >> 3) + m - 1]: four float4 LDS reads per frame
//   D  lane (b, j):     partial (i, f, g, o) of the unit for stream j; two permlane32 swaps + adds fold the K-halves
//                       and leave (i, g) in the low 32 lanes, (f, o) in the high ones — the S = 2 arrangement of
//                       LstmCell<32>: same activations, same exchange, c kept redundantly in both halves.
// One s_barrier per frame publishes h (LDS ring of 4 frames) to the four waves. Two helper waves ride the same barriers,
// one frame-step per barrier: wave P runs the pre pass (LPF -> pre-gain ramp -> EQ if pre) of the four streams as a
// continuous systolic cascade, a 16-lane row per stream, 6 frames ahead of the cell; wave Q computes Dense(H,1) of the
// frame the cell finished a step ago (row-parallel: 2 MACs per lane + a DPP row sum), skip / output gain, and feeds the
// post pass (DC blocker -> EQ if post -> master ramp), the same cascade form, whose last stage writes the block buffer
// in place. A frame-step of a helper is ~30-40 instructions; the cell's is the critical path.
//
// STATUS: an A/B partner, not a form the pool picks (AIDAX_KERNEL=q4 selects it; tests keep it correct). Measured on
// cfg2 (profiles/r03_cfg2_q4_ab.txt): the four cell waves alone run a block in 70.9 us against 76.6 us for the
// pipeline's recurrent waves alone — the MFMA form does shorten the frame — but the helper waves share SIMDs with two
// of the cell waves and every frame ends in a barrier all six waves must reach: with the pre-pass wave 74.7 us, with
// the Dense / post-pass wave 93.3 us, against 80.0 us for k_lstm_pipe, whose helper waves meet the recurrent wave once
// per 16 frames instead of every frame. What would make it pay is helper work off the cell's SIMDs, which a four-wave
// cell on a four-SIMD CU does not leave room for.
//
// Numerics: chain passes operation for operation as chain_step (bit-exact vs the oracle); the cell's dot products are
// summed in another order than k_lstm / k_lstm_pipe (two accumulator chains per K-half), so recurrent state agrees
// with them to ~1e-7, with the oracle within the usual bound (tests/test_gpu_q4.py).
#include <type_traits>

#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQ4Streams = 4;
constexpr int kQ4CellWaves = 4;
constexpr int kQ4Waves = kQ4CellWaves + 2;        // + P + Q
constexpr int kQ4Lead = 7;                        // steps the pre pass runs ahead of the cell: its cascade is <= 6 stages deep, and the
                                                  // cell fetches a frame's input a step early
constexpr int kQ4Tail = kQ4Lead + 2 + 5;          // steps after the last frame entered: cell, Dense (two steps behind the cell:
                                                  // its LDS operands are fetched a step ahead), post cascade drain
constexpr int kQ4HRing = 4;                       // frames of h in LDS
constexpr int kQ4XRing = 16;                      // frames of pre-pass output in LDS

__host__ __device__ constexpr int q4_row_stride(int H) { return H + 4; }
__host__ __device__ constexpr size_t q4_lds_floats(int H, int n_frames)
{
    return (size_t)kQ4Streams * ((n_frames + 3) & ~3) + (size_t)kQ4HRing * kQ4Streams * q4_row_stride(H) + kQ4XRing * kQ4Streams + 8;
}

__device__ __forceinline__ void q4_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// X = [xL | xH], Y = [yL | yH] over the wave's halves  ->  low lanes xL + xH, high lanes yL + yH
__device__ __forceinline__ float q4_fold(float x, float y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

__device__ __forceinline__ float q4_row_sum16(float v)
{
    v = v + dpp_take<0xB1, 0xf>(v);
    v = v + dpp_take<0x4E, 0xf>(v);
    v = v + dpp_take<0x141, 0xf>(v);
    v = v + dpp_take<0x140, 0xf>(v);
    return v;
}

// One step of a systolic cascade lane, as chain_step<.., true> (aidax_device.h): the lane works on sample s - stage.
// Returns true when it produced a sample (left in `carry`).
__device__ __forceinline__ bool q4_chain_step(ChainPass& c, int stage, bool run, float head, float& carry, int s, int n)
{
    const float from_left = dpp_row_shr1(carry);
    const float x = stage == 0 ? head : from_left;
    const int idx = s - stage;
    if (run && idx >= 0 && idx < n) {
        const double xd = x;                                // Biquad::process, Biquad.h:53-58
        const double yd = xd * c.a0 + c.z1;
        c.z1 = xd * c.a1 + c.z2 - c.b1 * yd;
        c.z2 = xd * c.a2 - c.b2 * yd;
        float y = c.active ? (float)yd : x;
        const float gm = c.g.next();
        if (stage == c.gain_lane) y = y * gm;
        carry = y;
        return true;
    }
    return false;
}

template <int H>
__global__ __launch_bounds__(kQ4Waves * kWave) void k_lstm_q4(LaunchArgs a)
{
    static_assert(H == 32, "8 units x 2 K-halves per wave, four waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HS = q4_row_stride(H);
    constexpr int KSTEPS = H / 2 + 1;                       // 17: x, then 16 columns of the half
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int nP = (n + 3) & ~3;
    const int s0 = (int)blockIdx.x * kQ4Streams;
    const int S = (int)a.n_streams;

    float* buf = smem;                                      // [4][nP]: the streams' blocks, input -> output in place
    float* hh  = buf + kQ4Streams * nP;                     // [ring][4][HS]
    float* xq  = hh + kQ4HRing * kQ4Streams * HS;           // [ring][4]: pre-pass output
    int*   flg = reinterpret_cast<int*>(xq + kQ4XRing * kQ4Streams);     // [4] live, [4] net in circuit

    if (n == 0) {                                           // pre-run (:607-609): targets latch, nothing else moves
        if (tid < kQ4Streams && s0 + tid < S) {
            const StreamCtl& ctl = a.ctl[s0 + tid];
            StreamState& st = a.st[s0 + tid];
            const uint32_t pending0 = st.pending;
            if (pending0 & PEND_ACTIVATE) { st.pre_mem = st.pre_tgt; st.master_mem = st.master_tgt; }
            st.pre_tgt = ctl.pre_target;
            st.pending = pending0 & ~PEND_ACTIVATE;
        }
        return;
    }

    // ---- the four blocks into LDS (one wave per row), the streams' flags
    if (wave < kQ4Streams) {
        float* row = buf + wave * nP;
        if (s0 + wave < S) load_block(row, a.in + (size_t)(s0 + wave) * n, n, lane);
        else for (int i = lane; i < n; i += kWave) row[i] = 0.f;
    } else if (wave == kQ4CellWaves) {
        for (int i = lane; i < kQ4XRing * kQ4Streams; i += kWave) xq[i] = 0.f;
        if (lane < kQ4Streams) {
            const bool valid = s0 + lane < S;
            const uint32_t f = valid ? a.ctl[s0 + lane].flags : 0u;
            const bool live = valid && (f & CTL_ENABLED);
            flg[lane] = live ? 1 : 0;
            flg[kQ4Streams + lane] = (live && (f & CTL_NET_ON)) ? 1 : 0;
        }
    } else {
        for (int i = lane; i < kQ4HRing * kQ4Streams * HS; i += kWave) hh[i] = 0.f;
    }
    __syncthreads();

    const int steps = n + kQ4Tail;

    if (wave < kQ4CellWaves) {
        // ------------------------------------------------------------------ the cell: 8 units x 2 K-halves x 4 streams
        const int b = lane >> 2, j = lane & 3;
        const int q = b >> 3, u = 8 * wave + (b & 7);
        const float* W = a.wpack + (size_t)wave * kQ4Regs * kWave + lane;
        float wr[KSTEPS];
#pragma unroll
        for (int m = 0; m < KSTEPS; ++m) wr[m] = W[m * kWave];
        const f32x4 cinit = { W[KSTEPS * kWave], W[(KSTEPS + 1) * kWave], W[(KSTEPS + 2) * kWave], W[(KSTEPS + 3) * kWave] };
        const bool hi = lane >= 32;                         // low half ends up with (i, g), high half with (f, o)
        const float ms = hi ? 0.5f : 1.f, ka = hi ? 0.5f : 1.f, kb = hi ? 0.5f : 0.f;     // tanh | sigmoid as ka*tanh(ms*v)+kb
        const bool upd = flg[kQ4Streams + j] != 0;          // a bypassed / disabled stream's model state does not move
        const int sg = s0 + j < S ? s0 + j : S - 1;
        float* nnst = a.nn + (size_t)sg * a.nn_stride;
        float c = upd ? nnst[H + u] : 0.f;
        float hcur = upd ? nnst[u] : 0.f;
        if (!hi) hh[((kQ4HRing - 1) * kQ4Streams + j) * HS + u] = hcur;      // h(-1): the slot "before" frame 0
        const float in_gain = a.in_gain;
        if (!(AIDAX_TUNE(a) & 1)) __builtin_amdgcn_s_setprio(3);
        // LDS addresses of the lane, per ring slot (the frame loop is unrolled over the four slots of the h ring)
        const float* hrd = hh + j * HS + 16 * q;            // + slot * 4 * HS: the 16 columns of h(t-1) this K-half contracts
        float* hwr = hh + j * HS + u;                       // + slot * 4 * HS: where h(t) of this lane's unit goes
        const float* xrd = xq + j;                          // + (t & 15) * 4
        const bool work = !(AIDAX_TUNE(a) & 64);
        // one frame; SLOT = t & 3 at compile time. `x` arrives prefetched (frame t), the next frame's is fetched here.
        auto frame = [&](int t, auto slot_c, float& x) {
            constexpr int SLOT = decltype(slot_c)::value, PREV = (SLOT + kQ4HRing - 1) & (kQ4HRing - 1);
            const f32x4* hv = reinterpret_cast<const f32x4*>(hrd + PREV * kQ4Streams * HS);
            const f32x4 h0 = hv[0], h1 = hv[1], h2 = hv[2], h3 = hv[3];
            const float xn = xrd[((t + 1) & (kQ4XRing - 1)) * kQ4Streams];      // written at least a step ago (kQ4Lead)
            __builtin_amdgcn_sched_barrier(0);               // all five reads in flight before the first MFMA waits for one
            f32x4 acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[0], x, cinit, 0, 0, 0);
            f32x4 acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[1], h0.x, f32x4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[2], h0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[3], h0.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4], h0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[5], h1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[6], h1.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[7], h1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[8], h1.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[9], h2.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[10], h2.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[11], h2.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[12], h2.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[13], h3.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[14], h3.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[15], h3.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[16], h3.w, acc0, 0, 0, 0);
            const f32x4 g = acc0 + acc1;
            const float v0 = q4_fold(g.x, g.y);             // low lanes: i, high lanes: f
            const float v1 = q4_fold(g.z, g.w);             // low lanes: g, high lanes: o
            const float a0 = fast_sigmoid(v0);
            const float a1 = __builtin_fmaf(tanh_rat(v1 * ms), ka, kb);
            const Pair p0 = share_halves(a0), p1 = share_halves(a1);
            const float cn = __builtin_fmaf(p0.hi, c, p0.lo * p1.lo);
            const float hn = p1.hi * tanh_rat(cn);
            if (upd) { c = cn; hcur = hn; }
            hwr[SLOT * kQ4Streams * HS] = hn;                // (both halves hold the unit's h: same address, same value)
            x = xn * in_gain;                               // out[i] *= input_gain
        };
        for (int s = 0; s < kQ4Lead; ++s) q4_barrier();
        float x = xrd[0] * in_gain;                         // frame 0's input
        int t = 0;
        if (work) {
            for (; t + 4 <= n; t += 4) {
                frame(t, std::integral_constant<int, 0>{}, x); q4_barrier();
                frame(t + 1, std::integral_constant<int, 1>{}, x); q4_barrier();
                frame(t + 2, std::integral_constant<int, 2>{}, x); q4_barrier();
                frame(t + 3, std::integral_constant<int, 3>{}, x); q4_barrier();
            }
            for (; t < n; ++t) {                            // ragged tail (t & 3 counts up from 0 again)
                switch (t & 3) {
                case 0: frame(t, std::integral_constant<int, 0>{}, x); break;
                case 1: frame(t, std::integral_constant<int, 1>{}, x); break;
                default: frame(t, std::integral_constant<int, 2>{}, x); break;
                }
                q4_barrier();
            }
        } else {
            for (; t < n; ++t) q4_barrier();
        }
        for (int s = 0; s < kQ4Tail - kQ4Lead; ++s) q4_barrier();
        if (upd && !hi && s0 + j < S) { nnst[u] = hcur; nnst[H + u] = c; }
    } else if (wave == kQ4CellWaves) {
        // ------------------------------------------------------------------ P: pre pass of the four streams, a row each
        const int j = lane >> 4, stage = lane & 15;
        const bool valid = s0 + j < S;
        const int sc = valid ? s0 + j : S - 1;
        const StreamCtl& ctl = a.ctl[sc];
        StreamState& st = a.st[sc];
        const uint32_t flags = ctl.flags;
        const uint32_t pending0 = st.pending;
        const bool live = flg[j] != 0;
        const bool net_on = flg[kQ4Streams + j] != 0;
        ChainPass cp{};
        cp.K = (flags & CTL_EQ_PRE) ? 6 : 1;
        cp.gain_lane = 0;
        const int k = stage < cp.K ? stage : 0;
        const int slot = pre_slot(k);
        const bool act = k == 0 ? (flags & CTL_LPF_ON) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        chain_load(cp, ctl, st, slot, act);
        const float pre_mem = (pending0 & PEND_ACTIVATE) ? st.pre_tgt : st.pre_mem;       // activate(): clearToTargetValue (:341-342)
        const float pre_tgt = ctl.pre_target;                                             // :513
        cp.g.arm(pre_mem, pre_tgt, ctl.pre_coef);
        const bool run = live && stage < cp.K;
        const bool writer = run && stage == cp.K - 1;
        const float* row = buf + j * nP;
        float carry = 0.f;
        float head_next = row[0];
        for (int s = 0; s < steps; ++s) {
            const float head = head_next;
            head_next = row[s + 1 < n ? s + 1 : n - 1];     // a step ahead: the read's latency is off the step's path
            if (!(AIDAX_TUNE(a) & (32 | 256)) && q4_chain_step(cp, stage, run, head, carry, s, n) && writer)
                xq[((s - stage) & (kQ4XRing - 1)) * kQ4Streams + j] = carry;
            q4_barrier();
        }
        if (run && cp.active) { st.z[slot][0] = cp.z1; st.z[slot][1] = cp.z2; }          // a bypassed biquad keeps its state (:622)
        if (valid && stage == 0) {
            uint32_t pending = pending0 & ~PEND_ACTIVATE;
            if (live) {
                st.pre_mem = cp.g.mem;
                if (net_on) {                                 // :634-640 (the targets follow the ports whatever the model reads)
                    float p_mem[2] = { st.p_mem[0], st.p_mem[1] }, p_tgt[2] = { st.p_tgt[0], st.p_tgt[1] }, p_step[2] = { st.p_step[0], st.p_step[1] };
#pragma unroll
                    for (int i = 0; i < 2; ++i) {             // LinearValueSmoother::setTargetValue (:209-216)
                        const float nt = ctl.p_target[i];
                        if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
                            p_tgt[i] = nt;
                            p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
                        }
                    }
                    if (pending & PEND_PARAM_FIRST) {         // paramFirstRun
                        pending &= ~PEND_PARAM_FIRST;
                        p_mem[0] = p_tgt[0];
                        p_mem[1] = p_tgt[1];
                    }
                    st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
                    st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
                    st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
                }
            } else if (pending0 & PEND_ACTIVATE) {            // hard bypass (:612-619): only the latches move
                st.pre_mem = st.pre_tgt;
                st.master_mem = st.master_tgt;
            }
            st.pre_tgt = pre_tgt;
            st.pending = pending;
        }
    } else {
        // ------------------------------------------------------------------ Q: Dense, skip / gain, post pass, a row each
        const int j = lane >> 4, stage = lane & 15;
        const bool valid = s0 + j < S;
        const int sc = valid ? s0 + j : S - 1;
        const StreamCtl& ctl = a.ctl[sc];
        StreamState& st = a.st[sc];
        const uint32_t flags = ctl.flags;
        const uint32_t pending0 = st.pending;
        const bool live = flg[j] != 0;
        const bool net_on = flg[kQ4Streams + j] != 0;
        ChainPass cp{};
        cp.K = (flags & CTL_EQ_POST) ? 6 : 1;
        cp.gain_lane = cp.K - 1;
        const int k = stage < cp.K ? stage : 0;
        const int slot = post_slot(k);
        const bool act = k == 0 ? (flags & CTL_DC_ON) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        chain_load(cp, ctl, st, slot, act);
        const float master_mem = (pending0 & PEND_ACTIVATE) ? st.master_tgt : st.master_mem;
        const float master_tgt = ctl.master_target;                                       // :654
        cp.g.arm(master_mem, master_tgt, ctl.master_coef);
        const bool run = live && stage < cp.K;
        const bool writer = run && stage == cp.K - 1;
        const float* wd = a.wpack + (size_t)kQ4CellWaves * kQ4Regs * kWave;               // Dense weights [H], bias
        const float wd0 = wd[2 * stage], wd1 = wd[2 * stage + 1], bd = wd[H];
        const float in_gain = a.in_gain, out_gain = a.out_gain;
        const bool skip = a.input_skip != 0;
        float* row = buf + j * nP;
        float carry = 0.f;
        float2 hv_next = { 0.f, 0.f };
        float xin_next = 0.f;
        for (int s = 0; s < steps; ++s) {
            const int tq = s - kQ4Lead - 2;                  // the frame the cell finished two steps ago: its h and the pre-pass
            float o = 0.f;                                   // output were fetched during the previous step
            const float2 hv = hv_next;
            const float xin = xin_next;
            {
                const int tn = tq + 1 < 0 ? 0 : tq + 1;      // next step's frame: h(tn) was published a step ago
                hv_next = *reinterpret_cast<const float2*>(hh + ((tn & (kQ4HRing - 1)) * kQ4Streams + j) * HS + 2 * stage);
                xin_next = xq[(tn & (kQ4XRing - 1)) * kQ4Streams + j];
            }
            if (tq >= 0 && tq < n && !(AIDAX_TUNE(a) & (32 | 128))) {
                const float part = __builtin_fmaf(wd1, hv.y, wd0 * hv.x);
                const float y = q4_row_sum16(part) + bd;
                const float xg = xin * in_gain;
                o = skip ? xg + y : y;                       // out[i] (+)= forward
                o = o * out_gain;                            // out[i] *= output_gain
                if (!net_on) o = xin;                        // :631-632: the model is not in circuit
            }
            if (!(AIDAX_TUNE(a) & (32 | 128)) && q4_chain_step(cp, stage, run, o, carry, tq, n) && writer) row[tq - stage] = carry;
            q4_barrier();
        }
        if (run && cp.active) { st.z[slot][0] = cp.z1; st.z[slot][1] = cp.z2; }
        if (valid && live && stage == cp.K - 1) { st.master_mem = cp.g.mem; st.master_tgt = master_tgt; }
    }

    // ---- results back to HBM (a disabled stream's row is still its input: the raw copy of :612-619)
    __syncthreads();
    if (wave < kQ4Streams && s0 + wave < S) store_block(a.out + (size_t)(s0 + wave) * n, buf + wave * nP, n, lane);
}

// ---------------------------------------------------------------- host side
bool q4_serves(int cell, int hidden, int input_size) { return cell == 0 && hidden == 32 && input_size == 1; }
size_t q4_lds_bytes(int hidden, uint32_t n_frames) { return q4_lds_floats(hidden, (int)n_frames) * sizeof(float); }

hipError_t launch_q4_kernel(int hidden, const LaunchArgs& a, hipStream_t stream)
{
    if (hidden != 32) return hipErrorInvalidValue;
    const size_t lds = q4_lds_bytes(hidden, a.n_frames);
    if (lds > 64 * 1024) return hipErrorInvalidValue;      // the pool uses this form for blocks up to kQ4MaxFrames
    const uint32_t groups = (a.n_streams + kQ4Streams - 1) / kQ4Streams;
    hipLaunchKernelGGL(k_lstm_q4<32>, dim3(groups), dim3(kQ4Waves * kWave), lds, stream, a);
    return hipGetLastError();
}

}  // namespace aidax
