// aidax_kernels.hip — gfx950 kernels of the rt-neural-generic hot path.
//
// One wavefront (64 lanes) per mono stream; one workgroup = one wavefront, so a
// pool of N streams is a grid of N workgroups (N >> 256 CUs fills the chip; the
// streams share nothing but the read-only weights, so there is no inter-
// workgroup traffic and no XCD-sensitive reuse to remap for).
//
// Per 256-frame block a stream moves 1 KiB in + 1 KiB out of HBM with one
// coalesced 16 B/lane access each; everything in between lives on chip:
//   * the audio block in LDS (read per sample as a wave-uniform broadcast),
//   * every weight of the recurrent layer in VGPRs (pre-shuffled by the host
//     into the lane mapping of aidax_layout.h),
//   * h exchanged through a 4*H-byte LDS line (ds_write_b32 + ds_read_b128
//     broadcasts), c in registers,
//   * biquad z1/z2 (fp64) and smoother memories in registers of the lanes
//     that own the stage.
//
// The 7-stage chain of run() (rt-neural-generic.cpp:621-659) is executed as
//   pre pass  : LPF -> pre-gain ramp -> (EQ if EQPOS==pre)   systolic over lanes
//   NN pass   : applyModel, strictly sequential in time
//   post pass : DC blocker -> (EQ if EQPOS==post) -> master ramp, systolic
// In a systolic pass lane k owns biquad k of the cascade and works on sample
// s-k at step s, taking its input from lane k-1 through a DPP row shift: the
// cascade's six fp64 recurrences advance in one instruction stream.
//
// Numerics: biquads are fp64 exactly as common/Biquad.h:53-58 (this file is
// built with -ffp-contract=off, so each line below is one IEEE operation, as in
// the reference built without FMA contraction): bit-exact vs the CPU oracle.
// The smoothers are the same fp32 recurrences as ValueSmoother.hpp. The NN is
// fp32 with explicit fmaf chains; sigmoid/tanh use v_exp_f32 + v_rcp_f32
// (abs error ~1e-7), compared to tolerance 1e-5 (rt-neural-generic.h:182).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cfloat>

#include <mutex>
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

// --------------------------------------------------------------- LSTM cell
// Lane mapping per aidax_layout.h. Gate index g = part + S*e; json column order
// i|f|c|o (pinned by the bundled goldens). The packer folds the activations' input scales into the rows
// (-log2 e for a sigmoid evaluated as 1/(1+2^v), 0.5 - exact - for one evaluated as 0.5*tanh(0.5 v)+0.5).
// S = 4: after the dot products every lane of a unit gets all four gate activations (permlane swaps) and c, h are
// kept redundantly. S = 2 (part 0: i, g; part 1: f, o): part 0 sends i*g over with ONE swap and the cell state lives in
// part 1 only (c, h of the part-0 lanes are scratch; they publish into a spare slot) — three instructions instead of six
// on the recurrence's critical path.
template <int H, bool ROTP = true>
struct LstmCell {
    static constexpr LaneMap M = lstm_map(H);
    static constexpr int S = M.S, SLOTS = M.slots, NU = M.NU, GPL = M.GPL;
    static constexpr int PACK = lstm_pack_regs(H);         // of the model's primary record (the Dense tail sits behind it)
    static constexpr int STATE = 2 * H;
    static constexpr bool ROT4 = ROTP && lstm_rot4(H);     // S = 4, H <= 16: every recurrent FMA by row rotation, no LDS read
    static constexpr int KW = lstm_row_weights(H, ROT4);   // recurrent weights per gate row in this instantiation's record

    float w[NU][GPL][KW];
    // ROT (H = 32): the two gate rows' weights for the sixteen units that come through LDS, SIDE BY SIDE — operand pairs of v_pk_fma_f32:
    // one instruction advances both rows' accumulators by one unit (see step)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 wl2[16];
    float wx[NU][GPL][kMaxInputs];
    float bias[NU][GPL];
    float wd[NU], bd;
    float aka[GPL], akb[GPL];
    float c[NU], h[NU];
    int part, slot;
#ifdef AIDAX_PIPE_TRACE
    unsigned long long ts6 = 0;
    unsigned long long ts[6];             // measurement build: s_memtime at five points of ONE frame of the unrolled stage (scratch/pipe_frame.py)
#define PT_STAMP(k) do { if constexpr (TR) { __builtin_amdgcn_sched_barrier(0); ts[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif
    static constexpr int kStatePart = S == 2 ? 1 : 0;      // the lanes whose c / h are the cell's
    // H = 32 (two full 16-lane rows of units): half of a row's recurrent FMAs take h(t-1) straight out of the
    // neighbours' registers — the lane's own unit plus 15 rotations within its row (v_fmac_f32_dpp) — while the reads of the
    // other 16 units are still on their way back from LDS: the write -> read turn-around of h, ~100 cycles per frame on
    // the recurrence's critical path, is covered by work, and four of the eight broadcast reads are gone. The packer
    // stores a lane's recurrent weights in the order they are used: [own row's units by rotation | the other row's].
    static constexpr bool ROT = ROTP && lstm_rot2(H);
    // where this instantiation's lane records start in the model's weight buffer (ROTP = false: the natural-order copy)
    static constexpr int kPackOffset = (!ROTP && lstm_has_alt_pack(H)) ? lstm_alt_pack_offset(H) : 0;

    __device__ __forceinline__ void load(const float* __restrict__ wbase, const float* __restrict__ st, int lane)
    {
        const float* __restrict__ wp = wbase + kPackOffset;
        part = lane / SLOTS;
        slot = lane % SLOTS;
        int r = 0;
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int e = 0; e < GPL; ++e) {
#pragma unroll
                for (int k = 0; k < KW; ++k) w[m][e][k] = wp[(r++) * kWave + lane];
#pragma unroll
                for (int i = 0; i < kMaxInputs; ++i) wx[m][e][i] = wp[(r++) * kWave + lane];
                bias[m][e] = wp[(r++) * kWave + lane];
            }
#pragma unroll
        for (int m = 0; m < NU; ++m) wd[m] = wp[(r++) * kWave + lane];
        bd = wp[(r++) * kWave + lane];
        if constexpr (ROTP && lstm_rot2(H)) {
            // (read once more from the record rather than copied out of w[][]: rows 0 and 1 of unit slot 0, KW + inputs + bias registers apart)
            constexpr int kRow = KW + kMaxInputs + 1;
#pragma unroll
            for (int k = 0; k < 16; ++k) wl2[k] = f32x2{ wp[(16 + k) * kWave + lane], wp[(kRow + 16 + k) * kWave + lane] };
        }
#pragma unroll
        for (int e = 0; e < GPL; ++e) {
            // row positions whose gate type differs between lanes are evaluated in the common
            // form ka*tanh(v)+kb: tanh (1,0), sigmoid (0.5,0.5) on a row the packer scaled by 0.5
            const bool is_tanh = (part + S * e) == 2;       // the candidate ("c") gate
            aka[e] = is_tanh ? 1.f : 0.5f;
            akb[e] = is_tanh ? 0.f : 0.5f;
        }
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            const int j = slot + m * SLOTS;
            h[m] = j < H ? st[j] : 0.f;
            c[m] = j < H ? st[H + j] : 0.f;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ st) const
    {
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            const int j = slot + m * SLOTS;
            if (part == kStatePart && j < H) { st[j] = h[m]; st[H + j] = c[m]; }
        }
    }

    // hbuf: H floats + at least one spare (rows of the kernels' LDS buffers are H + 4 long)
    __device__ __forceinline__ void publish_h(float* hbuf) const
    {
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            const int j = slot + m * SLOTS;
            if constexpr (S == 2 && !ROT) {
                if (j < H) hbuf[part == kStatePart ? j : H] = h[m];     // part 0's h is scratch: into the spare slot, no branch
            } else {
                if (j < H) hbuf[j] = h[m];   // the S lanes of a unit hold the same h: same address, same data
            }
        }
    }

    static constexpr int HID = H;

    // Dense(H,1) contribution of this lane for the h it holds (0 outside the state part / j >= H)
    __device__ __forceinline__ float dense_partial() const
    {
        float d = 0.f;
#pragma unroll
        for (int m = 0; m < NU; ++m) d = __builtin_fmaf(wd[m], h[m], d);
        return d;
    }

    // one sample: reads h(t-1) from hprev[0..H), leaves h(t) in registers and at hout[0..H).
    // NI = inputs that can be non-zero (the packer zero-fills the weights of absent inputs).
    // WINDOW > 0 fences the scheduler every WINDOW float4 loads of h, bounding the registers held by
    // in-flight LDS reads (the many-streams kernels trade that latency for occupancy); 0 = all at once.
    template <int NI = kMaxInputs, int WINDOW = 0, bool TR = false>
    __device__ __forceinline__ void step(float x0, float x1, float x2, const float* hprev, float* hout)
    {
        PT_STAMP(0);                                        // frame starts
        float acc[NU][GPL];
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int e = 0; e < GPL; ++e) {
                float a = bias[m][e];
                a = __builtin_fmaf(wx[m][e][0], x0, a);
                if constexpr (NI >= 2) a = __builtin_fmaf(wx[m][e][1], x1, a);
                if constexpr (NI >= 3) a = __builtin_fmaf(wx[m][e][2], x2, a);
                acc[m][e] = a;
            }
        if constexpr (ROT4) {
            // a 16-lane row is one gate of all the units: h(t-1) of every unit is a rotation away, nothing comes from LDS
            const float hr = h[0];
            float a0 = __builtin_fmaf(w[0][0][0], hr, acc[0][0]);
            fmac_row_ror_1_15(a0, hr, w[0][0]);               // rotations 1..15 in one asm statement (no s_nops in between)
            acc[0][0] = a0;
            (void)hprev;
        } else if constexpr (ROT) {
            // the other row's 16 units through LDS (issued first), this row's 16 out of the neighbours' registers
            const float4* hv = reinterpret_cast<const float4*>(hprev + 16 * (1 - ((slot >> 4) & 1)));
            const float4 q0 = hv[0], q1 = hv[1], q2 = hv[2], q3 = hv[3];
            __builtin_amdgcn_sched_barrier(0);
            const float hr = h[0];                          // h(t-1) of this lane's unit: every lane holds its unit's
            float a0 = acc[0][0], a1 = acc[0][1];
            a0 = __builtin_fmaf(w[0][0][0], hr, a0);
            a1 = __builtin_fmaf(w[0][1][0], hr, a1);
            fmac_row_ror_1_15(a0, hr, w[0][0]);               // rotations 1..15 of each row: one asm statement per row
            fmac_row_ror_1_15(a1, hr, w[0][1]);
            // (the rows' rotations as four chains — row x even / odd rotation, round robin — measured no better: 64.6 against 64.25 us)
            // (the four reads were issued ~35 instructions ago: ONE wait for all of them instead of one in front of each quad of FMAs)
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
            PT_STAMP(1);                                    // rotations issued, the other row's h has arrived
            // The sixteen units that came through LDS, both gate ROWS per instruction — {a0, a1} += {w0[k], w1[k]} * {h[k], h[k]}: v_pk_fma_f32 with
            // one half of the h pair selected for both results — on FOUR accumulator pairs in turn: a lone wave issues a packed FMA in the interval
            // of a plain one, but a DEPENDENT instruction only ~8.7 cycles after the one it waits for, an independent one after ~5.8
            // (scratch/ub/upk.hip). One pair chain of sixteen: 67.3 us (slower than the 32 plain FMAs on two chains, 66.2); four chains of four: 64.25.
            f32x2 pA = { a0, a1 }, pB = { 0.f, 0.f }, pC = { 0.f, 0.f }, pD = { 0.f, 0.f };
            const f32x2 hq[8] = { f32x2{ q0.x, q0.y }, f32x2{ q0.z, q0.w }, f32x2{ q1.x, q1.y }, f32x2{ q1.z, q1.w },
                                  f32x2{ q2.x, q2.y }, f32x2{ q2.z, q2.w }, f32x2{ q3.x, q3.y }, f32x2{ q3.z, q3.w } };
#pragma unroll
            for (int k2 = 0; k2 < 8; k2 += 2) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pA) : "v"(wl2[2 * k2]), "v"(hq[k2]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(pB) : "v"(wl2[2 * k2 + 1]), "v"(hq[k2]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pC) : "v"(wl2[2 * k2 + 2]), "v"(hq[k2 + 1]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(pD) : "v"(wl2[2 * k2 + 3]), "v"(hq[k2 + 1]));
            }
            const f32x2 a01 = (pA + pB) + (pC + pD);
            acc[0][0] = a01.x; acc[0][1] = a01.y;
            PT_STAMP(2);                                    // products done
        } else {
        const float4* hv = reinterpret_cast<const float4*>(hprev);
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 q = hv[k4];                        // same address in all lanes: broadcast
#pragma unroll
            for (int m = 0; m < NU; ++m)
#pragma unroll
                for (int e = 0; e < GPL; ++e) {
                    float a = acc[m][e];
                    a = __builtin_fmaf(w[m][e][4 * k4 + 0], q.x, a);
                    a = __builtin_fmaf(w[m][e][4 * k4 + 1], q.y, a);
                    a = __builtin_fmaf(w[m][e][4 * k4 + 2], q.z, a);
                    a = __builtin_fmaf(w[m][e][4 * k4 + 3], q.w, a);
                    acc[m][e] = a;
                }
            if constexpr (WINDOW > 0) { if ((k4 + 1) % WINDOW == 0) __builtin_amdgcn_sched_barrier(0); }
        }
        }
        __builtin_amdgcn_wave_barrier();                    // all reads of h(t-1) precede the publish below
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            float act[GPL];
#pragma unroll
            for (int e = 0; e < GPL; ++e) {
                if constexpr (S == 1) {                     // gate = e, known at compile time
                    act[e] = e == 2 ? tanh_rat(acc[m][e]) : sigmoid_pre(acc[m][e]);
                } else if (S == 2 && e == 0) {              // (i | f): sigmoid in both halves
                    act[e] = sigmoid_pre(acc[m][e]);
                } else {
                    act[e] = __builtin_fmaf(tanh_rat(acc[m][e]), aka[e], akb[e]);
                }
            }
            PT_STAMP(3);                                    // gate activations issued
            // c' = f*c + i*g, h = o*tanh(c') — where the four gates of a unit come together:
            float gi_gg, gf, go;
            if constexpr (S == 1) {                         // one lane holds all four
                gi_gg = act[0] * act[2]; gf = act[1]; go = act[3];
            } else if constexpr (S == 2) {
                // part 0 holds (i, g), part 1 (f, o): i*g crosses over with one swap (the register it lands in is a dead
                // accumulator), and the update happens where f and o are — in part 1; part 0's c and h are scratch
                const float ig = act[0] * act[1];
                const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, acc[m][1]), __builtin_bit_cast(unsigned, ig), false, false);
                const unsigned r0 = r[0];
                gi_gg = __builtin_bit_cast(float, r0); gf = act[0]; go = act[1];
            } else {
                // part q holds gate q: rows (v0,v1,v2,v3) -> (v0,v1,v0,v1),(v2,v3,v2,v3) -> each broadcast; every lane updates
                const Pair p = share_halves(act[0]);
                const Pair lo = share_rows(p.lo), hi = share_rows(p.hi);
                gi_gg = lo.lo * hi.lo; gf = lo.hi; go = hi.hi;
            }
            const float cn = __builtin_fmaf(gf, c[m], gi_gg);
            c[m] = cn;
#ifdef AIDAX_TANHC_EXP
            const float hn = go * tanh_exp(cn);             // experiment: the exp form for tanh(c) only (scratch/ab_tanhc.sh)
#else
            const float hn = go * tanh_rat(cn);
#endif
            if constexpr (ROT) h[m] = share_halves(hn).hi;  // the (f, o) half's h into both halves: the rotations read it
            else h[m] = hn;
        }
        PT_STAMP(4);                                        // h(t) in registers
        publish_h(hout);
        __builtin_amdgcn_wave_barrier();
        PT_STAMP(5);                                        // ... and on its way to LDS
    }
};

// ---------------------------------------------------------------- GRU cell
// A unit's z, r and candidate rows live on S lanes that split the recurrent dot products along K
// (S = 1: one lane, no exchange; S = 2: two half-length slices combined with one permlane swap per row).
// Keras reset_after form, json column order z|r|h, bias [2][3H]:
//   z = sig(Wz x + Uz h + bz0 + bz1), r likewise,
//   n = tanh(Wn x + bn0 + r*(Un h + bn1)),  h' = (1-z)*n + z*h
template <int H>
struct GruCell {
    static constexpr LaneMap M = gru_map(H);
    static constexpr int S = M.S, SLOTS = M.slots, NU = M.NU;
    static constexpr int KS = H / S;          // recurrent columns per lane (K-split across the S lanes of a unit)
    static constexpr int PACK = gru_pack_regs(H);
    static constexpr int STATE = H;
#ifdef AIDAX_PIPE_TRACE
    unsigned long long ts6 = 0, ts[6] = { 0, 0, 0, 0, 0, 0 };      // (LstmCell's frame stamps: not taken here)
#endif
    static_assert(KS % 4 == 0, "K slice must stay float4 aligned");

    float w[NU][3][KS];
    float wx[NU][3][kMaxInputs];   // z, r rows: part 0 only (they are summed across parts); candidate row: every part
    float bias[NU][3];             // z: b0+b1, r: b0+b1 (part 0 only), n: b0 (input side, every part)
    float bn1[NU];                 // n: b1 (recurrent side, part 0 only: it joins the summed partials)
    float wd[NU], bd;
    float h[NU];
    int part, slot;

    __device__ __forceinline__ void load(const float* __restrict__ wp, const float* __restrict__ st, int lane)
    {
        part = lane / SLOTS;
        slot = lane % SLOTS;
        int r = 0;
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
#pragma unroll
                for (int k = 0; k < KS; ++k) w[m][e][k] = wp[(r++) * kWave + lane];
#pragma unroll
                for (int i = 0; i < kMaxInputs; ++i) wx[m][e][i] = wp[(r++) * kWave + lane];
                bias[m][e] = wp[(r++) * kWave + lane];
            }
#pragma unroll
        for (int m = 0; m < NU; ++m) bn1[m] = wp[(r++) * kWave + lane];
#pragma unroll
        for (int m = 0; m < NU; ++m) wd[m] = wp[(r++) * kWave + lane];
        bd = wp[(r++) * kWave + lane];
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            const int j = slot + m * SLOTS;
            h[m] = j < H ? st[j] : 0.f;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ st) const
    {
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            const int j = slot + m * SLOTS;
            if (part == 0 && j < H) st[j] = h[m];
        }
    }

    __device__ __forceinline__ void publish_h(float* hbuf) const
    {
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            const int j = slot + m * SLOTS;
            if (j < H) hbuf[j] = h[m];       // the S lanes of a unit hold the same h
        }
    }

    static constexpr int HID = H;

    __device__ __forceinline__ float dense_partial() const
    {
        float d = 0.f;
#pragma unroll
        for (int m = 0; m < NU; ++m) d = __builtin_fmaf(wd[m], h[m], d);
        return d;
    }

    // sum of v over the S lanes of a unit, in every one of them
    __device__ __forceinline__ static float unit_sum(float v)
    {
        if constexpr (S == 1) return v;
        else if constexpr (S == 2) { const Pair p = share_halves(v); return p.lo + p.hi; }
        else { const Pair p = share_halves(v); const float t = p.lo + p.hi; const Pair q = share_rows(t); return q.lo + q.hi; }
    }

    template <int NI = kMaxInputs, int WINDOW = 0, bool TR = false>
    __device__ __forceinline__ void step(float x0, float x1, float x2, const float* hprev, float* hout)
    {
        float ax[NU][3], ar[NU][3];
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                float a = bias[m][e];
                a = __builtin_fmaf(wx[m][e][0], x0, a);
                if constexpr (NI >= 2) a = __builtin_fmaf(wx[m][e][1], x1, a);
                if constexpr (NI >= 3) a = __builtin_fmaf(wx[m][e][2], x2, a);
                ax[m][e] = a;
                ar[m][e] = e == 2 ? bn1[m] : 0.f;
            }
        const float4* hv = reinterpret_cast<const float4*>(hprev + part * KS);      // this lane's K slice
#pragma unroll
        for (int k4 = 0; k4 < KS / 4; ++k4) {
            const float4 q = hv[k4];
#pragma unroll
            for (int m = 0; m < NU; ++m)
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    float a = ar[m][e];
                    a = __builtin_fmaf(w[m][e][4 * k4 + 0], q.x, a);
                    a = __builtin_fmaf(w[m][e][4 * k4 + 1], q.y, a);
                    a = __builtin_fmaf(w[m][e][4 * k4 + 2], q.z, a);
                    a = __builtin_fmaf(w[m][e][4 * k4 + 3], q.w, a);
                    ar[m][e] = a;
                }
            if constexpr (WINDOW > 0) { if ((k4 + 1) % WINDOW == 0) __builtin_amdgcn_sched_barrier(0); }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < NU; ++m) {
            // partial sums of the S lanes -> full pre-activations in each of them
            const float zp = unit_sum(ax[m][0] + ar[m][0]);
            const float rp = unit_sum(ax[m][1] + ar[m][1]);
            const float nh = unit_sum(ar[m][2]);
            const float z = sigmoid_pre(zp);               // the packer scaled the z and r rows by -log2 e
            const float r = sigmoid_pre(rp);
            const float pre = __builtin_fmaf(r, nh, ax[m][2]);
            // The candidate through v_exp / v_rcp (the packer scaled its rows by 2 log2 e): four instructions against the
            // rational's sixteen. Its error is absolute (1.2e-7), which an LSTM's cell state integrates (why tanh_rat exists) —
            // a GRU does not: h = z h + (1 - z) n has the fixed point h = n, the error enters with weight 1 - z and leaves
            // at the same rate.
            const float n = tanh_exp_pre(pre);
            h[m] = __builtin_fmaf(z, h[m] - n, n);          // (1-z)*n + z*h
        }
        publish_h(hout);
        __builtin_amdgcn_wave_barrier();
    }
};

// ------------------------------------------------------------- the kernel
template <class Cell, bool HasCell>
__device__ __forceinline__ void stream_body(const LaunchArgs& a, float* smem)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.x;
    const int n = (int)a.n_frames;

    if constexpr (HasCell) {
        if (a.mode != MODE_CHAIN) {
            // ---- bare applyModel: warm-up over zeros, or explicit [n][I] inputs (stream 0)
            float* hbuf = smem;
            Cell cell;
            float* nnst = a.nn + (size_t)s * a.nn_stride;
            cell.load(a.wpack, nnst, lane);
            cell.publish_h(hbuf);
            __builtin_amdgcn_wave_barrier();
            StreamState* stp = a.st + s;
            float p1 = stp->p_mem[0], p2 = stp->p_mem[1];
            const float t1 = stp->p_tgt[0], t2 = stp->p_tgt[1], s1 = stp->p_step[0], s2 = stp->p_step[1];
            for (int t = 0; t < n; ++t) {
                float x = 0.f, q1 = 0.f, q2 = 0.f;
                if (a.mode == MODE_NN_ONLY) {
                    x = a.in[(size_t)t * a.input_size];
                    if (a.input_size >= 2) q1 = a.in[(size_t)t * a.input_size + 1];
                    if (a.input_size >= 3) q2 = a.in[(size_t)t * a.input_size + 2];
                } else {
                    if (a.input_size >= 2) q1 = lin_next(p1, t1, s1);
                    if (a.input_size >= 3) q2 = lin_next(p2, t2, s2);
                }
                x = x * a.in_gain;
                cell.step(x, q1, q2, hbuf, hbuf);
                const float y = wave_sum(cell.dense_partial()) + cell.bd;
                float o = a.input_skip ? x + y : y;
                o = o * a.out_gain;
                if (a.mode == MODE_NN_ONLY && lane == 0) a.out[t] = o;
            }
            cell.store(nnst);
            if (a.mode == MODE_WARMUP && lane == 0) { stp->p_mem[0] = p1; stp->p_mem[1] = p2; }
            return;
        }
    }

    // ---- MODE_CHAIN: run(), rt-neural-generic.cpp:489-518 + :607-659
    float* buf = smem;                                   // n_frames floats (rounded up to 4)
    float* hbuf = smem + ((n + 3) & ~3);
    const StreamCtl& ctl = a.ctl[s];
    StreamState& st = a.st[s];
    const uint32_t flags = ctl.flags;
    uint32_t pending = st.pending;

    float pre_mem = st.pre_mem, master_mem = st.master_mem;
    float pre_tgt = st.pre_tgt, master_tgt = st.master_tgt;
    if (pending & PEND_ACTIVATE) {                       // activate(): clearToTargetValue (:341-342)
        pre_mem = pre_tgt;
        master_mem = master_tgt;
        pending &= ~PEND_ACTIVATE;
    }
    pre_tgt = ctl.pre_target;                            // preGain.setTargetValue (:513), before every early-out

    const float* in_row = a.in + (size_t)s * n;
    float* out_row = a.out + (size_t)s * n;

    if (n == 0 || !(flags & CTL_ENABLED)) {              // pre-run (:607-609) / hard bypass (:612-619)
        if (n != 0 && out_row != in_row)
            for (int i = lane; i < n; i += kWave) out_row[i] = in_row[i];
        if (lane == 0) { st.pre_mem = pre_mem; st.master_mem = master_mem; st.pre_tgt = pre_tgt; st.pending = pending; }
        return;
    }

    load_block(buf, in_row, n, lane);
    __builtin_amdgcn_wave_barrier();

    // ---- pre pass: LPF (:622-626) -> pre-gain ramp (:627) -> EQ if pre (:628-630)
    {
        ChainPass c;
        const bool eq = flags & CTL_EQ_PRE;
        c.K = eq ? 6 : 1;
        c.gain_lane = 0;
        const int k = lane < c.K ? lane : 0;
        const int slot = pre_slot(k);
        bool act = k == 0 ? (flags & CTL_LPF_ON) != 0
                          : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        chain_load(c, ctl, st, slot, act);
        c.g.arm(pre_mem, pre_tgt, ctl.pre_coef);
        chain_run(c, buf, buf, n, lane);
        if (lane < c.K) { st.z[slot][0] = c.z1; st.z[slot][1] = c.z2; }
        pre_mem = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.g.mem), 0));
    }

    // ---- NN pass: applyModel (:631-644)
    if constexpr (HasCell) {
        if (flags & CTL_NET_ON) {
            float p_mem[2] = { st.p_mem[0], st.p_mem[1] };
            float p_tgt[2] = { st.p_tgt[0], st.p_tgt[1] };
            float p_step[2] = { st.p_step[0], st.p_step[1] };
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
            Cell cell;
            float* nnst = a.nn + (size_t)s * a.nn_stride;
            cell.load(a.wpack, nnst, lane);
            cell.publish_h(hbuf);
            __builtin_amdgcn_wave_barrier();
            const int I = a.input_size;
            for (int t = 0; t < n; ++t) {
                const float x = buf[t] * a.in_gain;       // out[i] *= input_gain
                float q1 = 0.f, q2 = 0.f;
                if (I >= 2) q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
                if (I >= 3) q2 = lin_next(p_mem[1], p_tgt[1], p_step[1]);
                cell.step(x, q1, q2, hbuf, hbuf);
                const float y = wave_sum(cell.dense_partial()) + cell.bd;
                float o = a.input_skip ? x + y : y;      // out[i] (+)= forward
                o = o * a.out_gain;                       // out[i] *= output_gain
                if (lane == 0) buf[t] = o;
                __builtin_amdgcn_wave_barrier();
            }
            cell.store(nnst);
            if (lane == 0) {
                st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
                st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
                st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
            }
        }
    }

    // ---- post pass: DC blocker (:645-650) -> EQ if post (:651-653) -> master ramp (:654-655)
    master_tgt = ctl.master_target;
    {
        ChainPass c;
        const bool eq = flags & CTL_EQ_POST;
        c.K = eq ? 6 : 1;
        c.gain_lane = c.K - 1;
        const int k = lane < c.K ? lane : 0;
        const int slot = post_slot(k);
        bool act = k == 0 ? (flags & CTL_DC_ON) != 0
                          : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        chain_load(c, ctl, st, slot, act);
        c.g.arm(master_mem, master_tgt, ctl.master_coef);
        chain_run(c, buf, buf, n, lane);
        if (lane < c.K) { st.z[slot][0] = c.z1; st.z[slot][1] = c.z2; }
        master_mem = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.g.mem), c.K - 1));
    }

    store_block(out_row, buf, n, lane);
    if (lane == 0) {
        st.pre_mem = pre_mem; st.master_mem = master_mem;
        st.pre_tgt = pre_tgt; st.master_tgt = master_tgt;
        st.pending = pending;
    }
}


// ======================================================================
// Wave-specialised pipeline (3 wavefronts per stream) — the low-stream-count form.
//
// Measured on MI355X (scratch/ubench.hip): a wavefront that is alone on its SIMD issues one
// VALU instruction per ~4.7 cycles, dependent or not; two or more resident waves share the
// SIMD at ~2.3 cycles per instruction. With <= ~1500 streams per GPU the one-wave-per-stream
// kernel therefore leaves half of every SIMD's issue slots empty, and everything that is not
// on the recurrence's critical path (the two biquad passes, smoothers, the Dense reduction,
// output gain, global I/O) only lengthens it. Here those parts move to sibling waves of the
// same workgroup, pipelined over 32-frame sub-blocks through LDS:
//
//   phase p:  wave P  pre pass + PARAM ramps of sub-block p      -> xq ring (3 deep)
//             wave N  recurrent cell over sub-block p-1           -> h history ring (2 x 32 rows)
//             wave Q  Dense + skip/out gain + post pass of p-2    -> global out
//   one workgroup barrier per phase; the N wave keeps weights, c and h in registers throughout.
// ======================================================================
#define AIDAX_STR2(x) #x
#define AIDAX_STR(x) AIDAX_STR2(x)
constexpr int kSB = 16;                 // frames per pipeline stage
constexpr int kStChainBlockP4 = 8;      // k_*_pipe4: frames per hand-over of the helper wave's cascades (two steps per stage)
constexpr int kRing = 2 * kSB;          // rows of the h history ring
constexpr int kPipeWaves = 3;
constexpr int kStage = 4 * kSB;         // floats of one hand-over stage of the input ring

__host__ __device__ constexpr int pipe_row_stride(int H) { return H + 4; }   // floats; keeps rows 16-B aligned
__host__ __device__ constexpr size_t pipe_lds_floats(int H, int n_frames)
{
    return (size_t)((n_frames + 3) & ~3) + 3 * kSB * 4 + (size_t)kRing * pipe_row_stride(H) + kSB + (size_t)(H + 4)
#ifdef AIDAX_PIPE_TRACE
         + 80                                               // the recurrent wave's stage stamps (scratch/pipe_frame.py)
#endif
        ;
}

// Wave P, a whole stage of a ONE-stage pre pass (no EQ in front of the model — the default): the stage's sixteen frames as one
// macro-step of the blocked form, twelve instructions per sample (fifteen while the gain ramp moves or the filter is out of
// circuit) instead of the systolic step's ~27 with its lane shift, selects and range logic. The helper waves share their SIMDs
// with other streams' recurrent waves, and for the small cells their instructions ARE the pass: LSTM-12 / GRU-8 at 1024
// streams spend as many issue slots on P and Q as on the cell. Same operations per sample as chain_step; lane 0 only.
template <bool PLAIN>
__device__ __forceinline__ void pre_stage_solo(ChainPass& c, const float* src, float* dst)
{
    static_assert(kSB % 4 == 0, "float4 moves");
    float v[kSB];
#pragma unroll
    for (int q = 0; q < kSB / 4; ++q) {
        const float4 t = reinterpret_cast<const float4*>(src)[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    const float g_fixed = c.g.mem;
#pragma unroll
    for (int i = 0; i < kSB; ++i) {
        const float x = v[i];
        const double xd = x;                                // Biquad::process, Biquad.h:53-58
        const double yd = xd * c.a0 + c.z1;
        c.z1 = xd * c.a1 + c.z2 - c.b1 * yd;
        c.z2 = xd * c.a2 - c.b2 * yd;
        if constexpr (PLAIN) {
            v[i] = (float)yd * g_fixed;
        } else {
            const float y = c.active ? (float)yd : x;
            v[i] = y * c.g.next();
        }
    }
#pragma unroll
    for (int q = 0; q < kSB / 4; ++q)
        reinterpret_cast<float4*>(dst)[q] = float4{ v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3] };
}

// The blocking host path of a ONE-stream pool (the LV2 instance's: aidax_pool_process on the pool's own page-locked block): the wave that
// has stored the block writes the sequence number the caller polls for — behind a system-scope fence, so the block is in the host's memory
// before the word is — and no packet has to follow the pass on its queue.
__device__ __forceinline__ void post_done_word(const LaunchArgs& a, int lane)
{
    if (a.done_word) {
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class Cell>
__device__ __forceinline__ void stream_body_pipe(const LaunchArgs& a, float* smem)
{
    constexpr int H = Cell::HID;
    constexpr int HS = pipe_row_stride(H);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = blockIdx.x;
    const int n = (int)a.n_frames;

#ifdef AIDAX_PIPE_TRACE
    const unsigned long long tr_w0 = wall_clock64(), tr_c0 = clock64();
#endif
    float*  inbuf = smem;                                                   // P private
    float*  xq    = smem + ((n + 3) & ~3);                                  // ring of 3 stages: x_pre[kSB] | p1[kSB] | p2[kSB] | -
    float*  hh    = xq + 3 * kStage;                                        // h history, kRing rows
    float*  qb    = hh + kRing * HS;                                        // Q private
    float*  wdl   = qb + kSB;                                               // Q private: Dense weights, natural order
#ifdef AIDAX_PIPE_TRACE
    uint32_t* stg_tr = reinterpret_cast<uint32_t*>(wdl + H + 4);            // [2 x 18 stages + 4]: wave N's clock at the begin / end of its work per pipeline step
#endif

    const StreamCtl& ctl = a.ctl[s];
    StreamState& st = a.st[s];
    const uint32_t flags = ctl.flags;
    const uint32_t pending0 = st.pending;
    const float* in_row = a.in + (size_t)s * n;
    float* out_row = a.out + (size_t)s * n;

    if (n == 0 || !(flags & CTL_ENABLED)) {              // pre-run (:607-609) / hard bypass (:612-619)
        if (n != 0 && out_row != in_row)
            for (int i = threadIdx.x; i < n; i += kPipeWaves * kWave) out_row[i] = in_row[i];
        if (threadIdx.x == 0) {
            if (pending0 & PEND_ACTIVATE) { st.pre_mem = st.pre_tgt; st.master_mem = st.master_tgt; }
            st.pre_tgt = ctl.pre_target;
            st.pending = pending0 & ~PEND_ACTIVATE;
        }
        if (a.done_word) {                                  // (every wave has copied a part)
            __threadfence_system();
            __syncthreads();
            post_done_word(a, (int)threadIdx.x);
        }
        return;
    }
    const bool net_on = (flags & CTL_NET_ON) != 0;
    const int I = a.input_size;
    const int n_sub = (n + kSB - 1) / kSB;

    // ---------------------------------------------------------------- role state
    ChainPass cp{};                       // P: pre pass / Q: post pass
    int slot = 0;
    float p_mem[2] = {0.f, 0.f}, p_tgt[2] = {0.f, 0.f}, p_step[2] = {0.f, 0.f};
    uint32_t pending = pending0 & ~PEND_ACTIVATE;
    float pre_tgt = 0.f, master_tgt = 0.f;
    Cell cell;
    float* nnst = a.nn + (size_t)s * a.nn_stride;
    float q_carry = 0.f;                  // Q: this lane's last cascade output (lane + 1 reads it by row shift), across stages
    int q_done = 0;                       // Q: samples of the block already stored
    double q_z1o = 0.0, q_z2o = 0.0;      // Q: the biquad state at the start of the block

    if (wave == 0) {                      // ---- P prologue
        load_block(inbuf, in_row, n, lane);
        const bool eq = flags & CTL_EQ_PRE;
        cp.K = eq ? 6 : 1;
        cp.gain_lane = 0;
        const int k = lane < cp.K ? lane : 0;
        slot = pre_slot(k);
        const bool act = k == 0 ? (flags & CTL_LPF_ON) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        chain_load(cp, ctl, st, slot, act);
        const float pre_mem = (pending0 & PEND_ACTIVATE) ? st.pre_tgt : st.pre_mem;
        pre_tgt = ctl.pre_target;
        cp.g.arm(pre_mem, pre_tgt, ctl.pre_coef);
        if (net_on) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                p_mem[i] = st.p_mem[i]; p_tgt[i] = st.p_tgt[i]; p_step[i] = st.p_step[i];
                const float nt = ctl.p_target[i];
                if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
                    p_tgt[i] = nt;
                    p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
                }
            }
            if (pending & PEND_PARAM_FIRST) {
                pending &= ~PEND_PARAM_FIRST;
                p_mem[0] = p_tgt[0];
                p_mem[1] = p_tgt[1];
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else if (wave == 1) {               // ---- N prologue
        // the recurrent wave is the critical path of the workgroup and of its SIMD, which it shares with helper
        // waves of other streams: it issues first (measured 95 -> 84 us per cfg2 block)
        if (!(AIDAX_TUNE(a) & 1)) __builtin_amdgcn_s_setprio(3);
        if (net_on) {
            cell.load(a.wpack, nnst, lane);
            cell.publish_h(hh + (kRing - 1) * HS);       // h(-1): the row "before" frame 0
        }
    } else {                              // ---- Q prologue
        const bool eq = flags & CTL_EQ_POST;
        cp.K = eq ? 6 : 1;
        cp.gain_lane = cp.K - 1;
        const int k = lane < cp.K ? lane : 0;
        slot = post_slot(k);
        const bool act = k == 0 ? (flags & CTL_DC_ON) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        chain_load(cp, ctl, st, slot, act);
        const float master_mem = (pending0 & PEND_ACTIVATE) ? st.master_tgt : st.master_mem;
        master_tgt = ctl.master_target;
        cp.g.arm(master_mem, master_tgt, ctl.master_coef);
        q_z1o = cp.z1; q_z2o = cp.z2;
        if (net_on) {
            const float* wd_nat = a.wpack + (size_t)Cell::PACK * kWave;     // [H] Dense weights then bias
            for (int i = lane; i < H + 1; i += kWave) wdl[i] = wd_nat[i];
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---------------------------------------------------------------- pipeline
    for (int p = 0; p < n_sub + 2; ++p) {
        if (wave == 0) {
            if (p < n_sub && !(AIDAX_TUNE(a) & 262144)) {       // (bit 262144, test build: wave P idle — what the helper waves cost the recurrent one; wrong output)
                const int base = p * kSB;
                const int cnt = n - base < kSB ? n - base : kSB;
                float* stage = xq + (p % 3) * kStage;
                if (cp.K == 1 && cnt == kSB) {
                    if (lane == 0) {
                        const double z1o = cp.z1, z2o = cp.z2;
                        if (cp.active && cp.g.mem * cp.g.coef + cp.g.tc == cp.g.mem) pre_stage_solo<true>(cp, inbuf + base, stage);
                        else pre_stage_solo<false>(cp, inbuf + base, stage);
                        if (!cp.active) { cp.z1 = z1o; cp.z2 = z2o; }       // a bypassed biquad keeps its state (:622)
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    chain_run<1>(cp, inbuf + base, stage, cnt, lane);
                }
                if (net_on && I >= 2) {
                    for (int t = 0; t < cnt; ++t) {
                        const float q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
                        const float q2 = I >= 3 ? lin_next(p_mem[1], p_tgt[1], p_step[1]) : 0.f;
                        if (lane == 0) { stage[kSB + t] = q1; stage[2 * kSB + t] = q2; }
                    }
                }
            }
        } else if (wave == 1) {
#ifdef AIDAX_PIPE_TRACE
            if (lane == 0 && p < 18) stg_tr[2 * p] = (uint32_t)(clock64() - tr_c0);
#endif
            if (net_on && p >= 1 && p <= n_sub) {
                const int base = (p - 1) * kSB;
                const int cnt = n - base < kSB ? n - base : kSB;
                const float* stage = xq + ((p - 1) % 3) * kStage;
                // rows of one sub-block are contiguous in the ring (base is a multiple of kSB)
                const float* hprev = hh + ((base + kRing - 1) & (kRing - 1)) * HS;
                float* hcur = hh + (base & (kRing - 1)) * HS;
                const float in_gain = a.in_gain;
                if (I == 1 && cnt == kSB) {
                    // snapshot models, whole stage: the stage's inputs travel into registers with four reads and the
                    // frame loop is unrolled, so a frame's only LDS traffic is h (one write, H/4 broadcast reads issued
                    // straight behind it) and nothing but the recurrence sits between two frames
                    float xr[kSB];
                    // The unrolled stage — ~1800 instructions issued by a lone wave, one per ~5 cycles — starts on a 64-byte boundary.
                    // Where it falls is otherwise an accident of everything in front of it in the code object: the shipped build of
                    // round 5 (the kernels' tune tests compiled out: a few instructions fewer in front) ran cfg2 at 68.5 - 68.9 us, the
                    // test build of the SAME loop at 66.2 - 66.5; with the boundary both run 66.4 - 66.7 (any start inside the line measures
                    // the same: profiles/r05_cfg2_alignment.txt).
                    asm volatile(".p2align 6");
#pragma unroll
                    for (int q = 0; q < kSB / 4; ++q) {
                        const float4 v = reinterpret_cast<const float4*>(stage)[q];
                        xr[4 * q] = v.x * in_gain; xr[4 * q + 1] = v.y * in_gain;       // out[i] *= input_gain
                        xr[4 * q + 2] = v.z * in_gain; xr[4 * q + 3] = v.w * in_gain;
                    }
                    // (round 6, measured and not kept: the sixteen multiplies on wave P — x * in_gain handed over in the stage's fourth quarter,
                    // read four frames at a time to stay under the register count of three waves per SIMD: 64.2 against 63.9 us)
#ifdef AIDAX_PIPE_TRACE
#pragma unroll
                    for (int t = 0; t < kSB; ++t) {
                        if (t == 8) cell.template step<1, 0, true>(xr[t], 0.f, 0.f, hcur + (t - 1) * HS, hcur + t * HS);
                        else cell.template step<1>(xr[t], 0.f, 0.f, t == 0 ? hprev : hcur + (t - 1) * HS, hcur + t * HS);
                        if (t == 9) cell.ts6 = __builtin_amdgcn_s_memtime();      // the frame after the traced one has been issued
                    }
#elif defined(AIDAX_PIPE_ROLL)
                    // (measurement build, scratch/r06_pipe_roll.sh: the stage as a LOOP of AIDAX_PIPE_ROLL frames per iteration instead of
                    // sixteen unrolled — 11 KB of straight-line code per stage against 0.7 KB per frame: does instruction fetch bound a lone wave?)
                    {
                        const float* hp = hprev;
                        float* hc = hcur;
#pragma unroll 1
                        for (int t0 = 0; t0 < kSB; t0 += AIDAX_PIPE_ROLL) {
                            float xs[AIDAX_PIPE_ROLL];
#pragma unroll
                            for (int i = 0; i < AIDAX_PIPE_ROLL; ++i) xs[i] = stage[t0 + i] * in_gain;
#pragma unroll
                            for (int i = 0; i < AIDAX_PIPE_ROLL; ++i) {
                                cell.template step<1>(xs[i], 0.f, 0.f, hp, hc);
                                hp = hc;
                                hc += HS;
                            }
                        }
                    }
#else
#pragma unroll
                    for (int t = 0; t < kSB; ++t)
                        cell.template step<1>(xr[t], 0.f, 0.f, t == 0 ? hprev : hcur + (t - 1) * HS, hcur + t * HS);
#endif
                } else if (I == 1) {                            // ragged last stage
                    for (int t = 0; t < cnt; ++t) {
                        cell.template step<1>(stage[t] * in_gain, 0.f, 0.f, hprev, hcur);
                        hprev = hcur;
                        hcur += HS;
                    }
                } else {
                    for (int t = 0; t < cnt; ++t) {
                        cell.template step<3>(stage[t] * in_gain, stage[kSB + t], I >= 3 ? stage[2 * kSB + t] : 0.f, hprev, hcur);
                        hprev = hcur;
                        hcur += HS;
                    }
                }
            }
        } else {
            if (p >= 2 && !(AIDAX_TUNE(a) & 524288)) {          // (bit 524288: wave Q idle)
                const int base = (p - 2) * kSB;
                const int cnt = n - base < kSB ? n - base : kSB;
                const int tl = lane < cnt ? lane : cnt - 1;
                const float xin = xq[((p - 2) % 3) * kStage + tl];
                float o = xin;
                if (net_on) {
                    asm volatile("" ::: "memory");          // do not hoist the Dense weights out of the phase loop
                    const float4* row = reinterpret_cast<const float4*>(hh + ((base + tl) & (kRing - 1)) * HS);
                    const float4* w4 = reinterpret_cast<const float4*>(wdl);
                    float y = wdl[H];                                       // Dense bias
#pragma unroll 2                                   // see k_nn: full unrolling costs 2H registers for the whole kernel
                    for (int k4 = 0; k4 < H / 4; ++k4) {
                        const float4 w = w4[k4];
                        const float4 hv = row[k4];
                        y = __builtin_fmaf(w.x, hv.x, y);
                        y = __builtin_fmaf(w.y, hv.y, y);
                        y = __builtin_fmaf(w.z, hv.z, y);
                        y = __builtin_fmaf(w.w, hv.w, y);
                    }
                    const float xg = xin * a.in_gain;
                    o = a.input_skip ? xg + y : y;                          // out[i] (+)= forward
                    o = o * a.out_gain;                                     // out[i] *= output_gain
                }
                if (lane < cnt) qb[lane] = o;
                __builtin_amdgcn_wave_barrier();
                // The post pass is ONE cascade over the whole block, not one per stage: lane k works on sample g - k at global
                // step g, whatever stage g falls in, so its K - 1 steps of fill and drain are paid once per block instead of once
                // per 16 frames, and every stage but the first and the last runs the steady-state step (no range tests). The
                // last cascade stage writes sample idx to inbuf[idx] — the input block, which wave P has long consumed that far
                // (it runs two stages ahead) — and what a stage completes leaves for HBM at its end.
                const bool first_stage = p == 2, last_stage = p == n_sub + 1;
                const bool q_run = lane < cp.K, q_writer = lane == cp.K - 1;
                // (round 6, measured and not kept: a ONE-stage post pass as lane-0 macro-steps like wave P's — twelve instructions per sample
                // instead of the systolic step's ~27, the same bits: 0.4 us SLOWER per cfg2 block at every placement tried, and sixteen values in
                // flight put the kernel over the register count of three waves per SIMD: profiles/r06_cfg2_pipe4.txt)
                if (first_stage) {                          // lanes k > 0 start k steps late: range tests
                    for (int t = 0; t < cnt; ++t) {
                        chain_step<1, true>(cp, lane, q_run, q_writer, qb[t], q_carry, inbuf, base + t, n);
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {                                    // every lane of the cascade has a sample in range: the lean step
                    float head_next = qb[0];
                    for (int t = 0; t < cnt; ++t) {
                        const float head = head_next;
                        head_next = qb[t + 1 < cnt ? t + 1 : t];
                        chain_step<1, false>(cp, lane, q_run, q_writer, head, q_carry, inbuf, base + t, n);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (last_stage)                             // the cascade's last K - 1 samples drain
                    for (int t = cnt; t < cnt + cp.K - 1; ++t) {
                        chain_step<1, true>(cp, lane, q_run, q_writer, 0.f, q_carry, inbuf, base + t, n);
                        __builtin_amdgcn_wave_barrier();
                    }
                // samples that have left the cascade: everything up to K - 1 behind the stage's last frame (all, at the end)
                const int hi = last_stage ? n : base + cnt - (cp.K - 1);
                for (int i = q_done + lane; i < hi; i += kWave) out_row[i] = inbuf[i];
                if (hi > q_done) q_done = hi;
            }
        }
#ifdef AIDAX_PIPE_TRACE
        if (wave == 1 && lane == 0 && p < 18) stg_tr[2 * p + 1] = (uint32_t)(clock64() - tr_c0);
#endif
        __syncthreads();
    }
#ifdef AIDAX_PIPE_TRACE
    if (wave == 1 && lane == 0) stg_tr[36] = (uint32_t)(clock64() - tr_c0);
#endif

    // ---------------------------------------------------------------- state write-back
    if (wave == 0) {
        if (lane < cp.K) { st.z[slot][0] = cp.z1; st.z[slot][1] = cp.z2; }
        if (lane == 0) {
            st.pre_mem = cp.g.mem;
            st.pre_tgt = pre_tgt;
            st.pending = pending;
            if (net_on) {
                st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
                st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
                st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
            }
        }
    } else if (wave == 1) {
        if (net_on) cell.store(nnst);
#ifdef AIDAX_PIPE_TRACE
        if constexpr (H == 32) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) out_row[8 + k] = __builtin_bit_cast(float, (uint32_t)(cell.ts[k] - cell.ts[0]));
                out_row[14] = __builtin_bit_cast(float, (uint32_t)(cell.ts6 - cell.ts[0]));
            }
            if (lane < 37) out_row[16 + lane] = __builtin_bit_cast(float, stg_tr[lane]);
        }
#endif
    } else {
        if (lane >= cp.K || !cp.active) { cp.z1 = q_z1o; cp.z2 = q_z2o; }      // a bypassed biquad keeps its state (:646)
        if (lane < cp.K) { st.z[slot][0] = cp.z1; st.z[slot][1] = cp.z2; }
        if (lane == cp.K - 1) { st.master_mem = cp.g.mem; st.master_tgt = master_tgt; }
        post_done_word(a, lane);                            // (this wave has stored the block)
#ifdef AIDAX_PIPE_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            out_row[0] = __builtin_bit_cast(float, (uint32_t)(tr_w0 & 0xffffffffu));
            out_row[1] = __builtin_bit_cast(float, (uint32_t)(wall_clock64() & 0xffffffffu));
            out_row[2] = __builtin_bit_cast(float, (uint32_t)(clock64() - tr_c0));
            out_row[3] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4));
            out_row[4] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20));
        }
#endif
    }
}

template <int H>
__global__ __launch_bounds__(kPipeWaves * kWave) void k_lstm_pipe(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body_pipe<LstmCell<H>>(a, smem);
}

template <int H>
__global__ __launch_bounds__(kPipeWaves * kWave) void k_gru_pipe(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body_pipe<GruCell<H>>(a, smem);
}


// ======================================================================
// The pipeline with ONE helper wave per FOUR streams (round 6, review item 4).
//
// k_*_pipe gives every stream two helper waves; a CU's four streams put eight of them on its four SIMDs, and what they issue there
// — two fp64 cascades and a Dense per stream and stage — costs the recurrent waves 3.3 us of cfg2's 64 (profiles/r05_cfg2_ledger.txt:
// a recurrent wave alone runs the launch in 61.3). Here a workgroup is a CU's four streams: four recurrent waves, one per SIMD, and
// ONE helper wave that carries all four streams' chain passes side by side — stream j's pre pass on lanes 12 j .. 12 j + 5, its post
// pass on lanes 12 j + 6 .. 12 j + 11 of the same fp64 instructions (k_conv_st's chain_macro_step, eight frames per hand-over, two
// steps per tick) — the four Dense(H, 1) + skip / gain tails on a lane per (stream, frame), the four rows' loads and stores. The
// helper issues per tick what ONE of the eight did, three SIMDs of four see no helper at all. A tick = a tile of sixteen frames:
//   tick p:  helper   Dense + skip / gain of tile p - d1 - 1, both passes' two macro-steps, tile p - d1 + 1 times in_gain
//            wave j   the recurrent cell of stream j over tile p - d1
// paced by progress words in LDS, not by barriers (see below). The host sends a pass here when the model takes the audio alone (no PARAM inputs),
// the block is whole tiles and every workgroup gets a CU of its own; streams that are disabled or have their model out of circuit ride along (their
// recurrent wave leaves at once, the helper copies or filters the row), the last workgroup may have fewer than four streams. k_*_pipe serves every
// other pass on the same state. Same operations per sample in the same order as k_*_pipe: bit-identical.
// ======================================================================
constexpr int kP4Streams = 4;
constexpr int kP4Helpers = 1;                          // helper waves; more than one take the ticks in turns (4: a recurrent wave meets a helper's burst every fourth tick — measured SLOWER, 65.2 against 64.6 us: profiles/r06_cfg2_pipe4.txt)
constexpr int kP4Waves = kP4Streams + kP4Helpers;
constexpr int kP4ChainLanes = 12 * kP4Streams;
constexpr int kP4HandFloats = 2 * kP4ChainLanes * kStChainBlockP4;
__host__ __device__ constexpr size_t pipe4_lds_floats(int H, int n_frames, bool conditioned)
{
    return (size_t)kP4Streams * n_frames + (size_t)kP4Streams * kRing * pipe_row_stride(H) + kP4HandFloats + (size_t)(H + 4) + 8 /* progress words */
         + 5 * kWave /* the cascades' state on its way from one helper wave to the next: z1, z2 (fp64), the ramp's memory */
         + (conditioned ? (size_t)2 * kP4Streams * n_frames : 0) /* PARAM1 / PARAM2 per frame (models with two or three inputs) */;
}
// a progress word in LDS, written by one wave and polled by others (LDS runs a wave's accesses in order: a word written behind the
// data it announces is seen behind it)
// (the words are named as LDS words — address space 3 — by hand: a volatile access through a generic pointer stays a FLAT instruction, which
// takes the long way to the LDS and is waited for with vmcnt(0), behind every global access the wave has in flight: 0.5 us of a cfg2 block)
typedef __attribute__((address_space(3))) int p4_lds_int;
typedef int p4_i2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) p4_i2 p4_lds_i2;
__device__ __forceinline__ int p4_peek(const int* w)
{
    const int v = *(const volatile p4_lds_int*)(const p4_lds_int*)(w);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void p4_post(int* w, int v, int lane)
{
    asm volatile("" ::: "memory");
    if (lane == 0) *(volatile p4_lds_int*)(p4_lds_int*)(w) = v;
    asm volatile("" ::: "memory");
}

// kCond: the model takes PARAM1 (and PARAM2) as inputs — a kernel of its own, so that the plain model's keeps its registers and its schedule
// (one body with the input count read at run time cost cfg2 2.5 us of 61.8: profiles/r06_pipe4_cells.txt)
template <class Cell, bool kCond>
__device__ __forceinline__ void stream_body_pipe4(const LaunchArgs& a, float* smem)
{
    constexpr int H = Cell::HID;
    constexpr int HS = pipe_row_stride(H);
    constexpr int B = kStChainBlockP4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s0 = kP4Streams * blockIdx.x;
    const int n = (int)a.n_frames;                        // a multiple of kSB (the host's promise)
    const int NT = n / kSB;
#ifdef AIDAX_P4_TRACE
    // measurement build (scratch/r06_p4_trace.sh): s_memrealtime (100 MHz) at the stations of one launch, written over the head of stream s0's output row
#define P4_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); tr[k] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    P4_STAMP(0);
#else
#define P4_STAMP(k) do { } while (0)
#endif
    float* rows = smem;                                   // [4][n]: a stream's block, processed in place by three actors a tile apart
    float* hh = rows + kP4Streams * n;                    // [4][kRing][HS]: h history
    float* hand = hh + kP4Streams * kRing * HS;           // the cascades' hand-over slots [2][48][B]
    float* wdl = hand + kP4HandFloats;                    // Dense weights, natural order, then the bias
    // Progress words instead of a barrier per tick: the recurrent waves never wait for EACH OTHER — four waves that leave a barrier
    // together run the same instructions in lockstep for the whole tile, and whether their LDS accesses and instruction fetches then
    // collide is decided by a few cycles of phase: the same kernel measured 61.4 .. 66.8 us over seven builds that differed by
    // 4 .. 64 bytes of padding (profiles/r06_cfg2_pipe4.txt). Left to themselves the waves drift apart and stay apart.
    int* prog = reinterpret_cast<int*>(wdl + H + 4);      // [0] tiles the helpers have made ready (pre pass done, times in_gain)
                                                          // [1] tiles whose Dense is done (their rows of the h ring are free)
                                                          // [2 + j] tiles the cell of stream j has finished
                                                          // [6] helper ticks done (whose turn it is)
    float* hstate = reinterpret_cast<float*>(prog + 8);   // [5][64]: a cascade lane's z1, z2, ramp memory between two helper waves
    float* prow = hstate + 5 * kWave;                     // [4 streams][2 params][n]: the smoothed PARAM inputs of every frame (kCond)
    const int I = kCond ? (int)a.input_size : 1;

    // every wave derives the launch's timing from the four control words (wave-uniform): the longest cascades set it
    int Kp = 1, Kq = 1;
    uint32_t fw[kP4Streams];                              // the four control words' flags (scalar loads: registers of the scalar file)
#pragma unroll
    for (int j = 0; j < kP4Streams; ++j) {
        // (the pool's last workgroup may have fewer than four streams: a missing one reads the last stream's word — an unconditional load of a
        // uniform address stays a scalar load, all four in flight together; behind a condition each became a loop of its own, one trip to memory
        // after the other in front of every wave's first instruction: 2 us of the launch)
        const int jc = s0 + j < (int)a.n_streams ? s0 + j : (int)a.n_streams - 1;
        const uint32_t f = a.ctl[jc].flags;
        fw[j] = s0 + j < (int)a.n_streams ? f : 0u;
        if ((f & CTL_EQ_PRE) && s0 + j < (int)a.n_streams) Kp = 6;
        if ((f & CTL_EQ_POST) && s0 + j < (int)a.n_streams) Kq = 6;
    }
    const int d1 = Kp / 2 + 1;                            // ticks between a tile entering the pre pass and the cell reading it
    const int T = (n / B - 1 + (Kq - 1) + 2 * (d1 + 1)) / 2 + 1;

    if (wave < kP4Streams) {
        // ---------------------------------------------------------------- a recurrent wave
        const int s = s0 + wave;
        // a stream that is not there, is disabled (:612-619) or has its model out of circuit (:631-632) has no cell to run: its wave says
        // "all tiles finished" and leaves (the helper copies or filters the row on its own)
        // (the weights and the state are requested FIRST, the control word behind them — one trip to memory, not two in a row: asked for in
        // front of the loads the word cost the launch 2 us; a stream out of circuit has read them for nothing)
        const int sc = s < (int)a.n_streams ? s : (int)a.n_streams - 1;
        float* nnst = a.nn + (size_t)sc * a.nn_stride;
        float* hj = hh + wave * kRing * HS;
        const float* row = rows + wave * n;
        Cell cell;
        if (!(AIDAX_TUNE(a) & 1)) __builtin_amdgcn_s_setprio(3);
        cell.load(a.wpack, nnst, lane);
        P4_STAMP(1);                                          // (the loads are issued)
        // (the word is one of the four every wave read with scalar loads above; and NOTHING in front of the first tile may depend on it but the
        // way out: with h(-1) published under the condition every vector load of the prologue was waited for there, and the first frames no
        // longer ran while the last weights were still arriving — 1.6 us of the launch)
        const uint32_t fl = wave == 0 ? fw[0] : wave == 1 ? fw[1] : wave == 2 ? fw[2] : fw[3];
        const bool net = (fl & (CTL_ENABLED | CTL_NET_ON)) == (CTL_ENABLED | CTL_NET_ON);
        cell.publish_h(hj + (kRing - 1) * HS);            // h(-1): the row "before" frame 0 (unconditionally: a stream out of circuit publishes into its own ring for nothing)
        __syncthreads();                                  // (the progress words are zero)
        P4_STAMP(2);
#ifndef AIDAX_P4_BARRIER
        if (!net) { p4_post(prog + 2 + wave, NT, lane); return; }
#endif
#ifdef AIDAX_P4_BARRIER
        // (measurement build, scratch/r06_p4_phase.sh: the first form — one workgroup barrier per tick, the waves in lockstep — with the
        // phase between the waves as a RUN-TIME parameter: bits 20 .. 23 of AIDAX_TUNE = sixteen-cycle steps the helper waits behind every
        // barrier, bits 24 .. 25 = steps recurrent wave j waits, times j. Same code for every setting: what changes is phase alone.)
        for (int tick = 0; tick < T; ++tick) {
            const int t = tick - d1;
            for (int i = 0; i < wave * ((AIDAX_TUNE(a) >> 24) & 3); ++i) asm volatile("s_nop 15");
            if (t >= 0 && t < NT) {
#else
        // (test build: bits 24 .. 27 of AIDAX_TUNE = sixteen-cycle steps recurrent wave j waits ONCE, times j, before its first tile — nothing
        // brings free-running waves back into step)
        for (int i = 0; i < wave * ((AIDAX_TUNE(a) >> 24) & 15); ++i) asm volatile("s_nop 15");
        // The two words a tile waits for are read AHEAD — requested four frames before the tile before ends, looked at when it has: the
        // helper is a tile ahead as a rule, the words only ever grow, and a stale look that already says "go" is as good as a fresh one. (Two
        // fresh looks per tile were two LDS round trips on the recurrent wave's critical path: 62.3 us for the recurrent waves alone against
        // 59.1 in the barrier form.)
        p4_i2 ahead = p4_i2{ 0, 0 };
        for (int t = 0; t < NT; ++t) {
            // the tile's inputs are ready, and the Dense has read the rows this tile overwrites (two tiles back)
            if (ahead.x <= t || ahead.y < t - 1)
                while (p4_peek(prog) <= t || p4_peek(prog + 1) < t - 1) __builtin_amdgcn_s_sleep(1);
#ifdef AIDAX_P4_TRACE
            if (t == 0) P4_STAMP(3);                          // (tile 0 is ready)
            if (t == 1) P4_STAMP(4);                          // (tile 0 is done and tile 1 ready)
            if (t == NT - 1) P4_STAMP(5);                     // (the last tile begins)
            if (wave == 0 && t < 32) {                        // (every tile's beginning, for the per-tile picture)
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (lane == 0) reinterpret_cast<unsigned long long*>(hstate)[8 + t] = now;
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            {
#endif
                const int base = t * kSB;
                const float* hprev = hj + ((base + kRing - 1) & (kRing - 1)) * HS;
                float* hcur = hj + (base & (kRing - 1)) * HS;
                const float* xs = row + base;             // x * in_gain (the helper's product, in place)
                float xr[kSB];
                asm volatile(".p2align 6");
                auto fetch_x = [&](int q) {
                    const float4 v = reinterpret_cast<const float4*>(xs)[q];
                    xr[4 * q] = v.x; xr[4 * q + 1] = v.y;
                    xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
                };
                fetch_x(0);
                if constexpr (!kCond) {
#pragma unroll
                    for (int f = 0; f < kSB; ++f) {
                        if ((f & 3) == 1 && f / 4 + 1 < kSB / 4) fetch_x(f / 4 + 1);
#ifndef AIDAX_P4_BARRIER
                        if (f == kSB - 4) ahead = *(const volatile p4_lds_i2*)(const p4_lds_i2*)(prog);
#endif
                        cell.template step<1>(xr[f], 0.f, 0.f, f == 0 ? hprev : hcur + (f - 1) * HS, hcur + f * HS);
                    }
                } else {
                    // conditioned models (input_size 2 / 3, :186-233): PARAM1 / PARAM2 of every frame are the helper's (LinearValueSmoother per sample)
                    const float* p1s = prow + (wave * 2 + 0) * n + base;
                    const float* p2s = prow + (wave * 2 + 1) * n + base;
                    float p1r[kSB], p2r[kSB];
                    auto fetch_p = [&](int q) {
                        const float4 u = reinterpret_cast<const float4*>(p1s)[q];
                        const float4 w = reinterpret_cast<const float4*>(p2s)[q];
                        p1r[4 * q] = u.x; p1r[4 * q + 1] = u.y; p1r[4 * q + 2] = u.z; p1r[4 * q + 3] = u.w;
                        p2r[4 * q] = w.x; p2r[4 * q + 1] = w.y; p2r[4 * q + 2] = w.z; p2r[4 * q + 3] = w.w;
                    };
                    fetch_p(0);
#pragma unroll
                    for (int f = 0; f < kSB; ++f) {
                        if ((f & 3) == 1 && f / 4 + 1 < kSB / 4) { fetch_x(f / 4 + 1); fetch_p(f / 4 + 1); }
#ifndef AIDAX_P4_BARRIER
                        if (f == kSB - 4) ahead = *(const volatile p4_lds_i2*)(const p4_lds_i2*)(prog);
#endif
                        cell.template step<3>(xr[f], p1r[f], I >= 3 ? p2r[f] : 0.f, f == 0 ? hprev : hcur + (f - 1) * HS, hcur + f * HS);
                    }
                }
            }
#ifdef AIDAX_P4_BARRIER
            __syncthreads();
#else
            p4_post(prog + 2 + wave, t + 1, lane);
#endif
        }
        P4_STAMP(6);                                          // (the last tile is done)
        cell.store(nnst);
#ifdef AIDAX_P4_TRACE
        P4_STAMP(7);
        {
            unsigned long long mine = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (lane == k) mine = tr[k];
            if (wave == 0 && lane < 8) reinterpret_cast<unsigned long long*>(hstate)[lane] = mine;      // (the helper takes them out)
        }
#endif
        return;
    }

    // ---------------------------------------------------------------- the helper wave(s)
    // kP4Helpers of them taking the ticks in turns: helper q runs ticks q, q + kP4Helpers, ... — every one with all the constants (the
    // cascades' coefficients, the Dense) of its own, the cascades' state (z1, z2, the ramp's memory: five words per lane) handed from
    // one to the next through LDS. (Four helpers, one per SIMD, so that each recurrent wave meets a burst every fourth tick instead of one
    // wave meeting all of them, measured no better than one: profiles/r06_cfg2_pipe4.txt.)
    const int hq = wave - kP4Streams;
    const int j = lane < kP4ChainLanes ? lane / 12 : kP4Streams - 1;      // this lane's stream as a cascade lane
    const int r = lane - 12 * (lane / 12);
    const bool isQ = r >= 6;
    const int stage = isQ ? r - 6 : r;
    const bool there = s0 + j < (int)a.n_streams;             // (the pool's last workgroup may have fewer than four streams: nothing of a missing one is read or written)
    const int sj = there ? s0 + j : (int)a.n_streams - 1;
    const StreamCtl& ctl = a.ctl[sj];
    StreamState& st = a.st[sj];
    const int slot = isQ ? post_slot(stage) : pre_slot(stage);
    // the four rows, dealt out over the helper waves (one 16-byte read per lane and 256 frames)
    // (all four requested before the first is waited for — the loop stays unrolled, a missing stream is a predicate, not a way out of the
    // loop: one after the other the four reads kept the recurrent waves waiting 2 us for their first tile)
    // (the workgroup's rows are ONE run of 4 n floats in the block and in LDS alike — [stream][n], dense: under 256 frames that is at most three
    // 16-byte reads per lane, all in flight together, whatever the block's length; a shorter block went stream by stream through the loop at the
    // end, four trips in a row: 0.6 - 0.7 us of a 64- or 128-frame launch, profiles/r06_cfg2_launch_stations.txt)
    // EVERYTHING the helper needs from memory is asked for here, in one go, before the first of it is put anywhere: the rows went into LDS as
    // they came, and behind a store through a generic pointer the compiler asks for nothing further until the store is done — five trips to
    // memory in a row (rows; Dense weights; coefficients and state; ramps; the Dense lanes' control word) where one is enough: 1.0 us in front
    // of EVERY launch's first tile (profiles/r06_cfg2_launch_stations.txt)
    float4 r0 = float4{0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, r3 = r0;
    if (kP4Helpers == 1 && n == 256) {
        // (a row per read: measured 0.15 us ahead of the general form below at this length — cfg2's)
        const bool t1 = s0 + 1 < (int)a.n_streams, t2 = s0 + 2 < (int)a.n_streams, t3 = s0 + 3 < (int)a.n_streams;      // (stream s0 is always there)
        r0 = reinterpret_cast<const float4*>(a.in + (size_t)s0 * n)[lane];
        r1 = reinterpret_cast<const float4*>(a.in + (size_t)(t1 ? s0 + 1 : s0) * n)[lane];
        r2 = reinterpret_cast<const float4*>(a.in + (size_t)(t2 ? s0 + 2 : s0) * n)[lane];
        r3 = reinterpret_cast<const float4*>(a.in + (size_t)(t3 ? s0 + 3 : s0) * n)[lane];
    } else if (kP4Helpers == 1 && n < 256) {
        const int have4 = ((int)a.n_streams - s0 < kP4Streams ? (int)a.n_streams - s0 : kP4Streams) * (n / 4);      // the 16-byte words that exist
        const float4* g4 = reinterpret_cast<const float4*>(a.in + (size_t)s0 * n);
        const int i0 = lane, i1 = kWave + lane, i2 = 2 * kWave + lane, i3 = 3 * kWave + lane;
        r0 = g4[i0 < have4 ? i0 : 0];
        r1 = g4[i1 < have4 ? i1 : 0];
        r2 = g4[i2 < have4 ? i2 : 0];
        r3 = g4[i3 < have4 ? i3 : 0];
    }
    static_assert(H + 1 <= kWave, "the Dense's weights and bias: a lane each");
    const float* wd_nat = a.wpack + (size_t)Cell::PACK * kWave;             // [H] Dense weights then bias
    const float wdv = wd_nat[lane < H + 1 ? lane : 0];
    ChainPass c;
    chain_load(c, ctl, st, slot, false);
    const uint32_t flags = ctl.flags;
    uint32_t pending = st.pending;
    float pre_mem = st.pre_mem, master_mem = st.master_mem, pre_tgt = st.pre_tgt, master_tgt = st.master_tgt;
    const float ctl_pre_target = ctl.pre_target, ctl_master_target = ctl.master_target, ctl_pre_coef = ctl.pre_coef, ctl_master_coef = ctl.master_coef;
    // the Dense's lanes: a lane per (stream, frame of the tile)
    const int dj = lane >> 4, df = lane & 15;
    const uint32_t fdj = a.ctl[s0 + dj < (int)a.n_streams ? s0 + dj : (int)a.n_streams - 1].flags;
    // the PARAM smoothers (LinearValueSmoother, ValueSmoother.hpp:166-241; run() :634-640) of a conditioned model: lane 48 + 2 j + i owns
    // PARAM(i + 1) of stream j — target changes and the first-run snap here (param_targets()' arithmetic), then a value per frame
    static_assert(kP4Helpers == 1, "the PARAM lanes' state is not handed from helper to helper");
    const bool plane = kCond && lane >= kP4ChainLanes && lane < kP4ChainLanes + 2 * kP4Streams;
    const int jp = plane ? (lane - kP4ChainLanes) >> 1 : 0, ip = lane & 1;
    const bool there_p = s0 + jp < (int)a.n_streams;
    const int sp = there_p ? s0 + jp : (int)a.n_streams - 1;
    float pm = 0.f, pt = 0.f, ps = 0.f, p_nt = 0.f, p_den = 1.f;
    uint32_t p_flags = 0, p_pending = 0;
    if constexpr (kCond) {
        pm = a.st[sp].p_mem[ip]; pt = a.st[sp].p_tgt[ip]; ps = a.st[sp].p_step[ip];
        p_nt = a.ctl[sp].p_target[ip]; p_den = a.ctl[sp].p_den;
        p_flags = a.ctl[sp].flags; p_pending = a.st[sp].pending;
    }
    // ---- ... and now it is put where it belongs
    if (kP4Helpers == 1 && n == 256) {
        reinterpret_cast<float4*>(rows)[lane] = r0;
        reinterpret_cast<float4*>(rows + n)[lane] = r1;
        reinterpret_cast<float4*>(rows + 2 * n)[lane] = r2;
        reinterpret_cast<float4*>(rows + 3 * n)[lane] = r3;
    } else if (kP4Helpers == 1 && n < 256) {
        float4* l4 = reinterpret_cast<float4*>(rows);
        const int i0 = lane, i1 = kWave + lane, i2 = 2 * kWave + lane, i3 = 3 * kWave + lane;
        if (i0 < n) l4[i0] = r0;
        if (i1 < n) l4[i1] = r1;
        if (i2 < n) l4[i2] = r2;
        if (i3 < n) l4[i3] = r3;
    } else {
        for (int jj = hq; jj < kP4Streams; jj += kP4Helpers) {
            if (s0 + jj >= (int)a.n_streams) break;
            const float4* src = reinterpret_cast<const float4*>(a.in + (size_t)(s0 + jj) * n);
            float4* dst = reinterpret_cast<float4*>(rows + jj * n);
            for (int i = lane; i < n / 4; i += kWave) dst[i] = src[i];
        }
    }
    if (hq == 0 && lane < H + 1) wdl[lane] = wdv;
    // this lane's stream: disabled = a raw copy, nothing advances but the latches (:612-619); model out of circuit = the chain alone (:631-632)
    const bool live = there && (flags & CTL_ENABLED) != 0;
    const bool netj = live && (flags & CTL_NET_ON) != 0;
    if (pending & PEND_ACTIVATE) { pre_mem = pre_tgt; master_mem = master_tgt; pending &= ~PEND_ACTIVATE; }     // activate(): :341-342
    pre_tgt = ctl_pre_target;
    if (live) master_tgt = ctl_master_target;
    const int Kpj = (flags & CTL_EQ_PRE) ? 6 : 1, Kqj = (flags & CTL_EQ_POST) ? 6 : 1;
    c.K = isQ ? Kqj : Kpj;
    c.gain_lane = isQ ? Kqj - 1 : 0;
    c.active = stage == 0 ? (flags & (isQ ? CTL_DC_ON : CTL_LPF_ON)) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
    c.g.arm(isQ ? master_mem : pre_mem, isQ ? master_tgt : pre_tgt, isQ ? ctl_master_coef : ctl_pre_coef);
    const bool run = lane < kP4ChainLanes && stage < c.K && live;
    const double z1o = c.z1, z2o = c.z2;
    ExpRamp g = c.g;
    const bool is_gain = stage == c.gain_lane;
    if (!is_gain) { g.mem = 1.f; g.coef = 1.f; g.tc = 0.f; }
    const bool last = stage == c.K - 1;
    const bool fussy = run && (!c.active || g.mem * g.coef + g.tc != g.mem);
    const bool plain = __builtin_amdgcn_ballot_w64(fussy) == 0;
    const int m0 = isQ ? 2 * (d1 + 1) : 0;
    float* myrow = rows + j * n;
    const float* hdj = hh + dj * kRing * HS;
    float* drow = rows + dj * n;
    const bool net_p = plane && there_p && (p_flags & (CTL_ENABLED | CTL_NET_ON)) == (CTL_ENABLED | CTL_NET_ON);
    if constexpr (kCond) {
        if (__builtin_fabsf(pt - p_nt) >= FLT_EPSILON) { pt = p_nt; ps = (pt - pm) / p_den; }      // setTargetValue (:209-216)
        if (p_pending & PEND_PARAM_FIRST) pm = pt;                                                // paramFirstRun (:636-640)
    }
    const bool pstep = net_p && ip < I - 1;                   // this PARAM is a model input: a smoothed value per frame
    float* pdst = prow + (jp * 2 + ip) * n;
    const bool netd = s0 + dj < (int)a.n_streams && (fdj & (CTL_ENABLED | CTL_NET_ON)) == (CTL_ENABLED | CTL_NET_ON);
    if (hq == 0 && lane < 8) prog[lane] = 0;
    P4_STAMP(1);                                              // (the helper's loads are issued)
    __syncthreads();
    P4_STAMP(2);
    for (int tick = hq; tick < T; tick += kP4Helpers) {
#ifdef AIDAX_P4_TRACE
        if (tick == 1) P4_STAMP(3);                           // (tick 0 done: tile 0 handed to the cells)
        if (tick == T - 1) P4_STAMP(4);                       // (the last tick begins)
#endif
        // my turn: the helper before me has finished tick - 1 and left the cascades' state
#ifdef AIDAX_P4_BARRIER
        for (int i = 0; i < ((AIDAX_TUNE(a) >> 20) & 15); ++i) asm volatile("s_nop 15");
#endif
        if (kP4Helpers > 1 && tick != 0) {
            while (p4_peek(prog + 6) < tick) __builtin_amdgcn_s_sleep(8);
            const double* zs = reinterpret_cast<const double*>(hstate);
            c.z1 = zs[lane]; c.z2 = zs[kWave + lane];
            g.mem = hstate[4 * kWave + lane];
        }
        // Dense + skip / output gain of the tile the cells finished a tick ago (applyModel :171-181), into the row: the post pass's input
        const int td = tick - d1 - 1;
        if (td >= 0 && td < NT) {
#ifndef AIDAX_P4_BARRIER
#pragma unroll
            for (int jj = 0; jj < kP4Streams; ++jj)
                while (p4_peek(prog + 2 + jj) <= td) __builtin_amdgcn_s_sleep(4);      // every cell has finished the tile
#endif
        }
        if (td >= 0 && td < NT && netd && !(AIDAX_TUNE(a) & 524288)) {      // (bit 524288, test build: no Dense — what it costs the recurrent wave it shares a SIMD with; wrong output)
            const int f = td * kSB + df;
            asm volatile("" ::: "memory");
            const float4* hrow = reinterpret_cast<const float4*>(hdj + (f & (kRing - 1)) * HS);
            const float4* w4 = reinterpret_cast<const float4*>(wdl);
            float y = wdl[H];
#pragma unroll 2
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const float4 w = w4[k4];
                const float4 hv = hrow[k4];
                y = __builtin_fmaf(w.x, hv.x, y);
                y = __builtin_fmaf(w.y, hv.y, y);
                y = __builtin_fmaf(w.z, hv.z, y);
                y = __builtin_fmaf(w.w, hv.w, y);
            }
            const float xg = drow[f];
            float o = a.input_skip ? xg + y : y;
            o = o * a.out_gain;
            drow[f] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (td >= 0 && td < NT) p4_post(prog + 1, td + 1, lane);
        if (!(AIDAX_TUNE(a) & 262144))                          // (bit 262144, test build: no chain passes; wrong output)
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            if (plain) chain_macro_step<true, B, kP4ChainLanes>(c, g, stage, run, last, myrow, hand, n / B, 2 * tick + hs - m0, lane);
            else chain_macro_step<false, B, kP4ChainLanes>(c, g, stage, run, last, myrow, hand, n / B, 2 * tick + hs - m0, lane);
        }
        // the tile the cells read next tick: times in_gain, in place (out[i] *= input_gain, :170; every stream's pre pass has finished it)
        const int ts = tick - d1 + 1;
        if (kCond && ts >= 0 && ts < NT) {
            if (pstep) {
#pragma unroll 4
                for (int f = 0; f < kSB; ++f) pdst[ts * kSB + f] = lin_next(pm, pt, ps);
            }
        }
        if (ts >= 0 && ts < NT) {
            const int f = ts * kSB + df;
            if (netd) drow[f] = drow[f] * a.in_gain;
            __builtin_amdgcn_wave_barrier();
            p4_post(prog, ts + 1, lane);
        }
#ifdef AIDAX_P4_BARRIER
        __syncthreads();
#endif
        if (kP4Helpers > 1 && tick + 1 < T) {
            // the next tick is another helper's: the state, then the word that says so
            double* zs = reinterpret_cast<double*>(hstate);
            zs[lane] = c.z1; zs[kWave + lane] = c.z2;
            hstate[4 * kWave + lane] = g.mem;
            __builtin_amdgcn_wave_barrier();
            p4_post(prog + 6, tick + 1, lane);
        }
    }
    P4_STAMP(5);                                              // (the last tick is done)
    if ((T - 1) % kP4Helpers != hq) return;               // the helper of the last tick holds the final state
    // ---------------------------------------------------------------- state write-back, the rows' stores
    if (is_gain) c.g = g;
    if (!run || !c.active) { c.z1 = z1o; c.z2 = z2o; }                      // a bypassed biquad keeps its state (:622, :646)
    if (run) { st.z[slot][0] = c.z1; st.z[slot][1] = c.z2; }
    // (a disabled stream: the gain memories as activate() left them, the pre-gain target latched, nothing else — :612-619 and k_*_pipe's early-out)
    if (there && lane < kP4ChainLanes && !isQ && stage == 0) { st.pre_mem = c.g.mem; st.pre_tgt = pre_tgt; }
    if (there && lane < kP4ChainLanes && isQ && stage == c.K - 1) { st.master_mem = c.g.mem; st.master_tgt = master_tgt; }
    if constexpr (kCond) {
        if (there && lane < kP4ChainLanes && r == 0) st.pending = netj ? (pending & ~PEND_PARAM_FIRST) : pending;      // (the first-run snap is the PARAM lanes')
        if (net_p) { a.st[sp].p_mem[ip] = pm; a.st[sp].p_tgt[ip] = pt; a.st[sp].p_step[ip] = ps; }
    } else {
        if (there && lane < kP4ChainLanes && r == 0) st.pending = netj ? param_targets(ctl, st, pending) : pending;      // run() :634-640 for a model without PARAM inputs
    }
    if (n == 256) {
        const float4 r0 = reinterpret_cast<const float4*>(rows)[lane], r1 = reinterpret_cast<const float4*>(rows + n)[lane];
        const float4 r2 = reinterpret_cast<const float4*>(rows + 2 * n)[lane], r3 = reinterpret_cast<const float4*>(rows + 3 * n)[lane];
        reinterpret_cast<float4*>(a.out + (size_t)s0 * n)[lane] = r0;
        if (s0 + 1 < (int)a.n_streams) reinterpret_cast<float4*>(a.out + (size_t)(s0 + 1) * n)[lane] = r1;
        if (s0 + 2 < (int)a.n_streams) reinterpret_cast<float4*>(a.out + (size_t)(s0 + 2) * n)[lane] = r2;
        if (s0 + 3 < (int)a.n_streams) reinterpret_cast<float4*>(a.out + (size_t)(s0 + 3) * n)[lane] = r3;
    } else if (kP4Helpers == 1 && n < 256) {
        const int have4 = ((int)a.n_streams - s0 < kP4Streams ? (int)a.n_streams - s0 : kP4Streams) * (n / 4);
        const float4* l4 = reinterpret_cast<const float4*>(rows);
        float4* o4 = reinterpret_cast<float4*>(a.out + (size_t)s0 * n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = k * kWave + lane;
            if (i < have4) o4[i] = l4[i];
        }
    } else {
        for (int jj = 0; jj < kP4Streams; ++jj) {
            if (s0 + jj >= (int)a.n_streams) break;
            float4* dst = reinterpret_cast<float4*>(a.out + (size_t)(s0 + jj) * n);
            const float4* src = reinterpret_cast<const float4*>(rows + jj * n);
            for (int i = lane; i < n / 4; i += kWave) dst[i] = src[i];
        }
    }
    post_done_word(a, lane);                                // (this wave has stored the block)
#ifdef AIDAX_P4_TRACE
    P4_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P4_STAMP(7);
    if (n >= 64) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out + (size_t)s0 * n);
        if (lane < 8) dst[lane] = reinterpret_cast<const unsigned long long*>(hstate)[lane];      // recurrent wave 0's
        unsigned long long mine = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (lane == k) mine = tr[k];
        if (lane < 8) dst[8 + lane] = mine;                                                       // the helper's
        if (lane < 32 && lane < NT && 16 + lane < n / 2) dst[16 + lane] = reinterpret_cast<const unsigned long long*>(hstate)[8 + lane];      // recurrent wave 0's tiles
    }
#endif
}

template <int H, bool kCond = false>
__global__ __launch_bounds__(kP4Waves * kWave) void k_lstm_pipe4(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body_pipe4<LstmCell<H>, kCond>(a, smem);
}

template <int H, bool kCond = false>
__global__ __launch_bounds__(kP4Waves * kWave) void k_gru_pipe4(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body_pipe4<GruCell<H>, kCond>(a, smem);
}

// ======================================================================
// Split form (3 launches) — the many-streams form.
//
// Once every SIMD holds several waves, the limit is instructions issued per sample, and the
// one-wave-per-stream kernel spends 112 of its ~300 instructions per sample on two systolic
// biquad passes that keep 1..6 of 64 lanes busy. HBM is nowhere near a limit (two extra trips
// of the audio block cost microseconds), so the chain is split into its own launches where
// one wave carries the chains of EIGHT streams side by side (8 lanes per stream, DPP row
// shifts never cross a stream's lane group), and the recurrent kernel keeps only the cell
// plus a lane-parallel Dense per 16 frames:
//     k_chain<true>   in  -> out : LPF -> pre-gain ramp -> EQ(pre)      8 streams / wave
//     k_nn<Cell>      out -> out : applyModel                           1 stream  / wave
//     k_chain<false>  out -> out : DC blocker -> EQ(post) -> master     8 streams / wave
// ======================================================================
constexpr int kChainStreams = 8;         // streams per wave in k_chain (8 lanes each)

template <bool PRE>
__global__ __launch_bounds__(kWave) void k_chain(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int n = (int)a.n_frames;
    const int nP = (n + 3) & ~3;
    // rows -> LDS, one coalesced row at a time (PRE reads the input block, POST works in place on out)
    for (int g = 0; g < kChainStreams; ++g) {
        const int s2 = blockIdx.x * kChainStreams + g;
        if (s2 < (int)a.n_streams && n != 0)
            load_block(smem + g * nP, (PRE ? a.in : a.out) + (size_t)s2 * n, n, lane);
    }
    __builtin_amdgcn_wave_barrier();
    chain_wave_pass<PRE>(a, blockIdx.x * kChainStreams, smem, nP, smem + kChainStreams * nP, n, lane, true, false);
    if (n != 0) {
        // PRE: every valid row goes to out (a disabled stream's row is still the raw input: the hard
        // bypass copy of :612-619); POST: only rows that were processed
        for (int g = 0; g < kChainStreams; ++g) {
            const int s2 = blockIdx.x * kChainStreams + g;
            if (s2 >= (int)a.n_streams) continue;
            const bool row_live = (a.ctl[s2].flags & CTL_ENABLED) != 0;
            if (PRE ? (row_live || a.out != a.in) : row_live)
                store_block(a.out + (size_t)s2 * n, smem + g * nP, n, lane);
        }
    }
}

constexpr int kNnSub = 16;               // frames between lane-parallel Dense passes in k_nn
constexpr int kNnRing = 2 * kNnSub;
constexpr int kNnWindow = 4;             // float4 h loads in flight per scheduling window (16 VGPRs)

__host__ __device__ constexpr size_t nn_lds_floats(int H, int n_frames)
{
    return (size_t)((n_frames + 3) & ~3) + (size_t)kNnRing * pipe_row_stride(H) + (size_t)(H + 4);
}

template <class Cell>
__global__ __launch_bounds__(kWave) void k_nn(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = Cell::HID;
    constexpr int HS = pipe_row_stride(H);
    const int lane = threadIdx.x;
    const int s = blockIdx.x;
    const int n = (int)a.n_frames;
    const StreamCtl& ctl = a.ctl[s];
    StreamState& st = a.st[s];
    const uint32_t flags = ctl.flags;
    if (n == 0 || !(flags & CTL_ENABLED) || !(flags & CTL_NET_ON)) return;      // :607-619, :631-632

    float* buf = smem;
    float* hh = smem + ((n + 3) & ~3);
    float* wdl = hh + kNnRing * HS;
    float* row = a.out + (size_t)s * n;
    load_block(buf, row, n, lane);
    const float* wd_nat = a.wpack + (size_t)Cell::PACK * kWave;
    for (int i = lane; i < H + 1; i += kWave) wdl[i] = wd_nat[i];

    float p_mem[2] = { st.p_mem[0], st.p_mem[1] };
    float p_tgt[2] = { st.p_tgt[0], st.p_tgt[1] };
    float p_step[2] = { st.p_step[0], st.p_step[1] };
    uint32_t pending = st.pending;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                    // LinearValueSmoother::setTargetValue (:209-216)
        const float nt = ctl.p_target[i];
        if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
            p_tgt[i] = nt;
            p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
        }
    }
    if (pending & PEND_PARAM_FIRST) {                // paramFirstRun (:636-640)
        pending &= ~PEND_PARAM_FIRST;
        p_mem[0] = p_tgt[0];
        p_mem[1] = p_tgt[1];
    }
    Cell cell;
    float* nnst = a.nn + (size_t)s * a.nn_stride;
    cell.load(a.wpack, nnst, lane);
    cell.publish_h(hh + (kNnRing - 1) * HS);
    __builtin_amdgcn_wave_barrier();

    const int I = a.input_size;
    const float in_gain = a.in_gain;
    for (int base = 0; base < n; base += kNnSub) {
        const int cnt = n - base < kNnSub ? n - base : kNnSub;
        const float* hprev = hh + ((base + kNnRing - 1) & (kNnRing - 1)) * HS;
        float* hcur = hh + (base & (kNnRing - 1)) * HS;
        if (I == 1) {
            float xin = buf[base];
            for (int t = 0; t < cnt; ++t) {
                const float x = xin * in_gain;
                xin = buf[base + t + 1 < n ? base + t + 1 : base + t];
                cell.template step<1, kNnWindow>(x, 0.f, 0.f, hprev, hcur);
                hprev = hcur;
                hcur += HS;
            }
        } else {
            for (int t = 0; t < cnt; ++t) {
                const float x = buf[base + t] * in_gain;
                const float q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
                const float q2 = I >= 3 ? lin_next(p_mem[1], p_tgt[1], p_step[1]) : 0.f;
                cell.template step<3, kNnWindow>(x, q1, q2, hprev, hcur);
                hprev = hcur;
                hcur += HS;
            }
        }
        // Dense(H,1) + skip + output gain for these frames, one lane per frame. The clobber keeps the
        // compiler from hoisting the H loop-invariant Dense weights into registers for the whole kernel.
        asm volatile("" ::: "memory");
        const int tl = lane < cnt ? lane : cnt - 1;
        const float4* hrow = reinterpret_cast<const float4*>(hh + ((base + tl) & (kNnRing - 1)) * HS);
        const float4* w4 = reinterpret_cast<const float4*>(wdl);
        float y = wdl[H];
#pragma unroll 2                       // fully unrolled this loop holds 2H registers of loads on top of the weights
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 w = w4[k4];
            const float4 hv = hrow[k4];
            y = __builtin_fmaf(w.x, hv.x, y);
            y = __builtin_fmaf(w.y, hv.y, y);
            y = __builtin_fmaf(w.z, hv.z, y);
            y = __builtin_fmaf(w.w, hv.w, y);
        }
        const float xg = buf[base + tl] * in_gain;
        float o = a.input_skip ? xg + y : y;
        o = o * a.out_gain;
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) buf[base + lane] = o;
        __builtin_amdgcn_wave_barrier();
    }
    store_block(row, buf, n, lane);
    cell.store(nnst);
    if (lane == 0) {
        st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
        st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
        st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
        st.pending = pending;
    }
}

struct NoCell { static constexpr int PACK = 0, STATE = 0; };

template <int H>
__global__ __launch_bounds__(kWave) void k_lstm(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body<LstmCell<H>, true>(a, smem);
}

template <int H>
__global__ __launch_bounds__(kWave) void k_gru(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body<GruCell<H>, true>(a, smem);
}

__global__ __launch_bounds__(kWave) void k_nomodel(LaunchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stream_body<NoCell, false>(a, smem);
}

// one-shot pokes from the control thread (activate / loading are host-latched in ctl;
// this sets StreamState.pending bits stream-ordered with the process launches)
__global__ void k_keep_warm() {}
hipError_t launch_keep_warm_kernel(int workgroups, hipStream_t stream)
{
    hipLaunchKernelGGL(k_keep_warm, dim3(workgroups), dim3(kWave), 0, stream);
    return hipGetLastError();
}

__global__ void k_set_pending(StreamState* st, uint32_t n_streams, int32_t stream, uint32_t bits)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    if (stream < 0 || (uint32_t)stream == i) st[i].pending |= bits;
}

// work_response(): fresh DynamicModel per stream (:1046-1061) — zero recurrent
// state, param smoothers re-created around the inherited targets.
__global__ void k_reset_for_model(StreamState* st, float* nn, uint32_t n_streams, uint32_t nn_stride, float p_den)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams) return;
    for (uint32_t k = 0; k < nn_stride; ++k) nn[(size_t)i * nn_stride + k] = 0.f;
    StreamState& s = st[i];
    for (int k = 0; k < 2; ++k) {
        const float old = s.p_tgt[k];
        // ctor zeros; setTargetValue(old) computes step from mem = 0; clearToTargetValue
        s.p_step[k] = __builtin_fabsf(0.f - old) >= FLT_EPSILON ? (old - 0.f) / p_den : 0.f;
        s.p_tgt[k] = __builtin_fabsf(0.f - old) >= FLT_EPSILON ? old : 0.f;
        s.p_mem[k] = s.p_tgt[k];
    }
    s.pending |= PEND_PARAM_FIRST;
}

// instantiate() (:283-321) for `n` streams: preGain target 1 cleared, masterGain target 0 cleared, biquad
// z = 0, PARAM smoothers zero, nothing pending.
__global__ void k_init_streams(StreamState* st, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StreamState s{};
    s.pre_mem = 1.f; s.pre_tgt = 1.f;
    st[i] = s;
}

// work() reads the PARAM targets of the playing model (:822-825) for the model it is about to build. `live` is
// being written by the audio side's passes while this runs on the worker's stream — the reference's worker reads
// the audio thread's smoothers just as unsynchronised. Each target is ONE relaxed device-scope 4-byte load (served by
// L2, never a torn or cached value): it is the target some run() before or during the prepare has set, which is all the
// reference promises; the first run() after the swap sets the targets from the ports again and snaps (:634-640).
__global__ void k_stage_params(const StreamState* live, StreamState* staged, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t0 = __hip_atomic_load(&live[i].p_tgt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float t1 = __hip_atomic_load(&live[i].p_tgt[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t* w = reinterpret_cast<uint32_t*>(&staged[i]);                  // an all-zero record but for the two targets
    for (uint32_t k = 0; k < sizeof(StreamState) / 4; ++k) w[k] = 0u;
    staged[i].p_tgt[0] = t0;
    staged[i].p_tgt[1] = t1;
}

// work_response(): the per-stream members of the new DynamicModel (PARAM smoothers as prepared and warmed up,
// paramFirstRun = true, :1053-1061) replace the old model's, stream-ordered with the passes.
__global__ void k_install_params(StreamState* live, const StreamState* staged, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 2; ++k) {
        live[i].p_mem[k] = staged[i].p_mem[k];
        live[i].p_step[k] = staged[i].p_step[k];
        live[i].p_tgt[k] = staged[i].p_tgt[k];
    }
    live[i].pending |= PEND_PARAM_FIRST;
}

// A fresh stream that continues another one's plugin instance (hub mode: an instance moves to the hub of its new model
// file): the DynamicModel it is about to get is built around the PARAM targets of the model that plays now (:822-825).
__global__ void k_set_param_targets(StreamState* st, float t0, float t1)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->p_tgt[0] = t0; st->p_tgt[1] = t1; }
}

// ... and at the swap (work_response, :868-875) the plugin's own DSP members stay what they are: the seven biquads'
// z1 / z2 and both gain smoothers (rt-neural-generic.h:311-317) travel from the old stream's record to the new one's.
// The PARAM smoothers belong to the new model and stay; activate() was not called by a swap.
__global__ void k_adopt_dsp(StreamState* dst, const StreamState* src)
{
    const int i = threadIdx.x;
    if (i < BQ_COUNT * 2) (&dst->z[0][0])[i] = (&src->z[0][0])[i];
    if (i == 0) {
        dst->pre_mem = src->pre_mem; dst->master_mem = src->master_mem;
        dst->pre_tgt = src->pre_tgt; dst->master_tgt = src->master_tgt;
        dst->pending &= ~PEND_ACTIVATE;
    }
}

// ------------------------------------------------------------ host dispatch
#define AIDAX_LSTM(H) { 0, H, k_lstm<H>, k_lstm_pipe<H>, k_nn<LstmCell<H, false>>, LstmCell<H>::PACK, LstmCell<H>::STATE, "k_lstm<" #H ">", "k_lstm_pipe<" #H ">", "k_chain+k_nn<lstm" #H ">", nullptr, "-", nullptr }
// ... with the four-streams-per-workgroup pipeline as well (BASELINE cfg2's cell)
#define AIDAX_LSTM_P4(H) { 0, H, k_lstm<H>, k_lstm_pipe<H>, k_nn<LstmCell<H, false>>, LstmCell<H>::PACK, LstmCell<H>::STATE, "k_lstm<" #H ">", "k_lstm_pipe<" #H ">", "k_chain+k_nn<lstm" #H ">", k_lstm_pipe4<H>, "k_lstm_pipe4<" #H ">", k_lstm_pipe4<H, true> }
// LSTM-64 / LSTM-80: the helper waves' share of the register file (pipe) resp. the Dense ring (split, H = 80) push the
// 4H-row cell past 512 registers — those forms would spill, so they do not exist; the pool serves these cells
// with k_quad / k_mfma, or the one-wave kernel when neither fits (pools with long blocks).
#define AIDAX_LSTM_WIDE(H, NN) { 0, H, k_lstm<H>, nullptr, NN, LstmCell<H>::PACK, LstmCell<H>::STATE, "k_lstm<" #H ">", "-", "k_chain+k_nn<lstm" #H ">", nullptr, "-", nullptr }
#define AIDAX_GRU(H)  { 1, H, k_gru<H>,  k_gru_pipe<H>,  k_nn<GruCell<H>>,  GruCell<H>::PACK,  GruCell<H>::STATE,  "k_gru<" #H ">", "k_gru_pipe<" #H ">", "k_chain+k_nn<gru" #H ">", nullptr, "-", nullptr }
#define AIDAX_LSTM_P4C(H) { 0, H, k_lstm<H>, k_lstm_pipe<H>, k_nn<LstmCell<H, false>>, LstmCell<H>::PACK, LstmCell<H>::STATE, "k_lstm<" #H ">", "k_lstm_pipe<" #H ">", "k_chain+k_nn<lstm" #H ">", nullptr, "k_lstm_pipe4<" #H ">", k_lstm_pipe4<H, true> }
#define AIDAX_GRU_P4C(H)  { 1, H, k_gru<H>,  k_gru_pipe<H>,  k_nn<GruCell<H>>,  GruCell<H>::PACK,  GruCell<H>::STATE,  "k_gru<" #H ">", "k_gru_pipe<" #H ">", "k_chain+k_nn<gru" #H ">", nullptr, "k_gru_pipe4<" #H ">", k_gru_pipe4<H, true> }
#define AIDAX_GRU_P4(H)  { 1, H, k_gru<H>,  k_gru_pipe<H>,  k_nn<GruCell<H>>,  GruCell<H>::PACK,  GruCell<H>::STATE,  "k_gru<" #H ">", "k_gru_pipe<" #H ">", "k_chain+k_nn<gru" #H ">", k_gru_pipe4<H>, "k_gru_pipe4<" #H ">", k_gru_pipe4<H, true> }

static const KernelEntry kTable[] = {
    // the 18 (cell, hidden) pairs of variant/generate_variant_hpp.py:4-6; input size is a run-time argument
    // (k_*_pipe4 where it measured ahead of k_*_pipe at a full pool of 1024 streams, profiles/r06_pipe4_cells.txt: LSTM-8 / 12 / 16 by 14 - 19 %,
    // GRU-8 / 12 / 16 by 9 - 14 %, LSTM-32 by 3 %; re-measured on the round's final kernel: GRU-24 by 6 %, LSTM-20 by 1 - 2 %; LSTM-24 and
    // GRU-20 / 32 measure 1 - 10 % behind and keep the three-wave pipeline.
    // A CONDITIONED model — PARAM1 / PARAM2 as inputs — runs k_*_pipe4<H, true> at every cell up to 32: 9 - 25 % ahead, same file)
#ifdef AIDAX_P4_ALL_CELLS      // (measurement build, scratch/r06_pipe4_cells.py: the plain models of every cell up to 32 on k_*_pipe4)
    AIDAX_LSTM_P4(8), AIDAX_LSTM_P4(12), AIDAX_LSTM_P4(16), AIDAX_LSTM_P4(20), AIDAX_LSTM_P4(24),
    AIDAX_LSTM_P4(32), AIDAX_LSTM(40), AIDAX_LSTM_WIDE(64, k_nn<LstmCell<64>>), AIDAX_LSTM_WIDE(80, nullptr),
    AIDAX_GRU_P4(8), AIDAX_GRU_P4(12), AIDAX_GRU_P4(16), AIDAX_GRU_P4(20), AIDAX_GRU_P4(24),
    AIDAX_GRU_P4(32), AIDAX_GRU(40), AIDAX_GRU(64), AIDAX_GRU(80),
#else
    AIDAX_LSTM_P4(8), AIDAX_LSTM_P4(12), AIDAX_LSTM_P4(16), AIDAX_LSTM_P4(20), AIDAX_LSTM_P4C(24),
    AIDAX_LSTM_P4(32), AIDAX_LSTM(40), AIDAX_LSTM_WIDE(64, k_nn<LstmCell<64>>), AIDAX_LSTM_WIDE(80, nullptr),
    AIDAX_GRU_P4(8), AIDAX_GRU_P4(12), AIDAX_GRU_P4(16), AIDAX_GRU_P4C(20), AIDAX_GRU_P4(24),
    AIDAX_GRU_P4C(32), AIDAX_GRU(40), AIDAX_GRU(64), AIDAX_GRU(80),
#endif
};

const KernelEntry* find_kernel(int cell, int hidden)
{
    for (const auto& e : kTable)
        if (e.cell == cell && e.hidden == hidden) return &e;
    return nullptr;
}

hipError_t launch_stream_kernel(const KernelEntry* e, const LaunchArgs& a, size_t lds_bytes, hipStream_t stream)
{
    void (*fn)(LaunchArgs) = e ? e->fn : k_nomodel;
    hipLaunchKernelGGL(fn, dim3(a.n_streams), dim3(kWave), lds_bytes, stream, a);
    return hipGetLastError();
}

size_t chain_lds_bytes(uint32_t n_frames)
{
    return ((size_t)kChainStreams * ((n_frames + 3) & ~3u) + kChainPackHandFloats) * sizeof(float);   // rows + hand-over slots
}

hipError_t launch_chain_pass(bool pre, const LaunchArgs& a, hipStream_t stream)
{
    const uint32_t groups = (a.n_streams + kChainStreams - 1) / kChainStreams;
    const size_t lds = chain_lds_bytes(a.n_frames);
    if (lds > kChainLdsLimit) return hipErrorInvalidValue;     // callers pick another form for such blocks
    if (lds > 64 * 1024) {
        const void* fn = pre ? reinterpret_cast<const void*>(k_chain<true>) : reinterpret_cast<const void*>(k_chain<false>);
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (pre) hipLaunchKernelGGL(k_chain<true>, dim3(groups), dim3(kWave), lds, stream, a);
    else hipLaunchKernelGGL(k_chain<false>, dim3(groups), dim3(kWave), lds, stream, a);
    return hipGetLastError();
}

// Split form: pre chain (8 streams/wave) -> recurrent cell -> post chain, in place on `out`
hipError_t launch_split_kernels(const KernelEntry* e, const LaunchArgs& a, hipStream_t stream)
{
    hipError_t err = launch_chain_pass(true, a, stream);
    if (err != hipSuccess) return err;
    if (e && !e->fn_nn) return hipErrorInvalidDeviceFunction;
    if (e && a.n_frames != 0)
        hipLaunchKernelGGL(e->fn_nn, dim3(a.n_streams), dim3(kWave), nn_lds_floats(e->hidden, (int)a.n_frames) * sizeof(float), stream, a);
    return launch_chain_pass(false, a, stream);
}

// Is the split form the better many-streams form for this cell? Yes when the lean recurrent kernel still
// gets >= 2 waves per SIMD (measured: LSTM-16 1.56x, LSTM-32 1.40x at 16384 streams) or at least the
// occupancy of the one-wave-per-stream kernel, and does not spill where that one does not. The widest
// cells (GRU-64: 254 vs 350 registers) lose their second wave per SIMD and stay on the one-wave form.
bool split_form_pays(const KernelEntry* e, uint32_t n_frames)
{
    if (!e->fn_nn) return false;
    int occ_wave = 0, occ_nn = 0;
    const size_t lds_wave = ((size_t)((n_frames + 3) & ~3u) + (size_t)e->hidden + 4) * sizeof(float);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_wave, e->fn, kWave, lds_wave) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_nn, e->fn_nn, kWave, nn_lds_floats(e->hidden, (int)n_frames) * sizeof(float)) != hipSuccess) return false;
    hipFuncAttributes aw{}, an{};
    if (hipFuncGetAttributes(&aw, reinterpret_cast<const void*>(e->fn)) != hipSuccess) return false;
    if (hipFuncGetAttributes(&an, reinterpret_cast<const void*>(e->fn_nn)) != hipSuccess) return false;
    if (an.localSizeBytes > aw.localSizeBytes) return false;
    return occ_nn >= occ_wave || occ_nn >= 8;          // blocks of one wave per CU: 8 = two waves per SIMD
}

size_t pipe_lds_bytes(int hidden, uint32_t n_frames) { return pipe_lds_floats(hidden, (int)n_frames) * sizeof(float); }

hipError_t launch_pipe_kernel(const KernelEntry* e, const LaunchArgs& a, hipStream_t stream, hipEvent_t done)
{
    if (!e->fn_pipe) return hipErrorInvalidDeviceFunction;
    if (done) {                                             // (the dispatch packet's completion signal is the event: see launch_pipe4_kernel)
        hipExtLaunchKernelGGL(e->fn_pipe, dim3(a.n_streams), dim3(kPipeWaves * kWave), pipe_lds_bytes(e->hidden, a.n_frames), stream, nullptr, done, 0, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(e->fn_pipe, dim3(a.n_streams), dim3(kPipeWaves * kWave), pipe_lds_bytes(e->hidden, a.n_frames), stream, a);
    return hipGetLastError();
}

// k_*_pipe4's workgroup asks for more than half a CU's LDS whatever its rows need: the dispatcher then cannot put two of them on one CU
// while another CU stands empty (a CU with two runs both at half speed, and the launch ends with them)
size_t pipe4_lds_bytes(int hidden, uint32_t n_frames, int input_size)
{
    const size_t need = pipe4_lds_floats(hidden, (int)n_frames, input_size > 1) * sizeof(float);
    return need > 81 * 1024 ? need : (size_t)81 * 1024;
}
hipError_t launch_pipe4_kernel(const KernelEntry* e, const LaunchArgs& a, hipStream_t stream, hipEvent_t done)
{
    void (*const kernel)(LaunchArgs) = a.input_size > 1 ? e->fn_pipe4c : e->fn_pipe4;      // (the conditioned model's kernel: PARAM rows beside the audio)
    if (!kernel || a.n_frames % kSB || a.n_frames == 0) return hipErrorInvalidValue;
    const size_t lds = pipe4_lds_bytes(e->hidden, a.n_frames, (int)a.input_size);
    {
        // (per kernel, once: more than 64 KiB of dynamic LDS has to be asked for)
        static std::mutex mu;
        static const void* raised[64];
        static int n_raised = 0;
        std::lock_guard<std::mutex> g(mu);
        const void* fn = reinterpret_cast<const void*>(kernel);
        bool done = false;
        for (int i = 0; i < n_raised; ++i) done = done || raised[i] == fn;
        if (!done) {
            const hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (err != hipSuccess) return err;
            if (n_raised < 64) raised[n_raised++] = fn;
        }
    }
    if (done) {
        // the dispatch packet's own completion signal is the event: no marker packet behind the pass (profiles/r06_host_pipeline.txt)
        hipExtLaunchKernelGGL(kernel, dim3((a.n_streams + kP4Streams - 1) / kP4Streams), dim3(kP4Waves * kWave), lds, stream, nullptr, done, 0, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kernel, dim3((a.n_streams + kP4Streams - 1) / kP4Streams), dim3(kP4Waves * kWave), lds, stream, a);
    return hipGetLastError();
}

// Streams whose 3-wave workgroups are all resident at once on this device; beyond that the
// one-wave-per-stream kernel has the better throughput (helper waves hold the N wave's VGPR budget).
int pipe_resident_streams(const KernelEntry* e, uint32_t n_frames, int device)
{
    if (!e->fn_pipe) return 0;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, e->fn_pipe, kPipeWaves * kWave,
                                                     pipe_lds_bytes(e->hidden, n_frames)) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu * cus;
}

hipError_t launch_set_pending(StreamState* st, uint32_t n_streams, int32_t stream, uint32_t bits, hipStream_t q)
{
    hipLaunchKernelGGL(k_set_pending, dim3((n_streams + 255) / 256), dim3(256), 0, q, st, n_streams, stream, bits);
    return hipGetLastError();
}

hipError_t launch_init_streams(StreamState* st, uint32_t n, hipStream_t q)
{
    hipLaunchKernelGGL(k_init_streams, dim3((n + 255) / 256), dim3(256), 0, q, st, n);
    return hipGetLastError();
}

hipError_t launch_stage_params(const StreamState* live, StreamState* staged, uint32_t n, hipStream_t q)
{
    hipLaunchKernelGGL(k_stage_params, dim3((n + 255) / 256), dim3(256), 0, q, live, staged, n);
    return hipGetLastError();
}

hipError_t launch_install_params(StreamState* live, const StreamState* staged, uint32_t n, hipStream_t q)
{
    hipLaunchKernelGGL(k_install_params, dim3((n + 255) / 256), dim3(256), 0, q, live, staged, n);
    return hipGetLastError();
}

hipError_t launch_set_param_targets(StreamState* st, float t0, float t1, hipStream_t q)
{
    hipLaunchKernelGGL(k_set_param_targets, dim3(1), dim3(64), 0, q, st, t0, t1);
    return hipGetLastError();
}

hipError_t launch_adopt_dsp(StreamState* dst, const StreamState* src, hipStream_t q)
{
    hipLaunchKernelGGL(k_adopt_dsp, dim3(1), dim3(64), 0, q, dst, src);
    return hipGetLastError();
}

hipError_t launch_reset_for_model(StreamState* st, float* nn, uint32_t n_streams, uint32_t nn_stride, float p_den, hipStream_t q)
{
    hipLaunchKernelGGL(k_reset_for_model, dim3((n_streams + 255) / 256), dim3(256), 0, q, st, nn, n_streams, nn_stride, p_den);
    return hipGetLastError();
}

}  // namespace aidax
