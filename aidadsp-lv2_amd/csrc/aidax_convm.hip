// aidax_convm.hip — k_conv_mfma: the causal dilated conv1d stack (BASELINE config #4) on the matrix cores.
//
// A conv layer over a block of frames is a contraction: out[frame][cout] = sum over (tap, cin) of
// x[cin][frame - (ksize-1-tap)*dilation] * K[tap][cin][cout] — [frames x ksize*Cin] . [ksize*Cin x Cout],
// 256 x 48 x 16 per layer for the 16-channel, 3-tap model. One workgroup (4 waves) per stream:
//
//   A (16 frames x 4 k)  read straight out of the channel-major LDS activation plane: lane supplies
//                        x[cin][t0 + (lane&15) - shift(tap)] — 16 consecutive frames per cin, and a plane
//                        stride of 16 (mod 64) floats puts the four cin groups of a fragment on disjoint banks;
//   B (4 k x 16 cout)    the layer's kernel as ready fragments (aidax_pack.cpp), staged in LDS once per layer
//                        together with a per-lane table of A-fragment plane offsets (no integer division
//                        in the loop);
//   D                    lane holds 4 consecutive frames of one output channel: tanh, one ds_write_b128.
//
// ONE activation plane, updated IN PLACE: a layer's sixteen frame tiles (four per wave) are accumulated in registers
// over all k-steps, every wave is done reading the plane at a barrier, and only then the outputs overwrite the block
// part of the plane and the next layer's history prefix its history part. That halves the LDS of a stream (34 KiB
// for the 16-channel model) — four workgroups per CU instead of two, all 1024 workgroups of BASELINE cfg4 resident at
// once — and takes one of the three barriers out of a layer. Per-layer input history ((ksize-1)*dilation frames per
// input channel) persists in HBM in the same layout as k_conv, so the two kernels are interchangeable mid-stream.
// Two instantiations: k_conv_mfma<false> is applyModel only (warm-up, self-test, and MODE_CHAIN between two packed
// k_chain launches — the form for pools with more streams than stay resident in one round); k_conv_mfma<true> is the
// whole run() of its stream, chain passes included (see below). 256 frames per launch; a pool made for longer host
// blocks sends them through in time slices (LaunchArgs::row_stride).
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kConvmThreads = 256;
constexpr int kConvmTiles = 4;            // frame tiles per wave per pass (256 frames = 16 tiles = 4 waves x 4)

// (convm_plane_stride: aidax_layout.h — the packer needs it for the full-block records)
__host__ __device__ inline size_t convm_lds_floats(const ConvDesc& d, int n_frames)
{
    const size_t stage = 2 * (size_t)d.max_k_steps * kWave;                    /* B fragments + A offsets; the fused form's  */
    return (size_t)d.channels * convm_plane_stride(d.max_hist, n_frames)       /* chain passes borrow it for their hand-over */
         + (stage > (size_t)kChainHandFloats ? stage : (size_t)kChainHandFloats)
         + 16 + 4                                                             /* Dense weights + bias      */
         + 8;                                                                 /* fused form: what the chain prologue leaves for the epilogue */
}

// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains vmcnt, i.e. it would wait for
// the register prefetches below and for the acknowledgement of every history store; threads of this kernel
// never exchange data through global memory.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// FUSED: the whole run() of the stream in this one launch (MODE_CHAIN only) — one wave of the workgroup runs the
// pre pass (LPF, pre-gain, EQ-pre) on the audio row where layer 0 will read it, and the post pass (DC blocker, EQ-post,
// master) on the row the Dense left, both in the blocked systolic form of k_chain. The chain is serial per stream
// (~9 + ~10 us for 256 frames on a wave of its own), but two launches, two round trips of the block through HBM and
// their gaps go away: BASELINE cfg4 86.4 -> 81.3 us. The four workgroups that share a CU start in step and run their
// chain passes at the same time, each on its wave 0: the dispatcher rotates the SIMD a workgroup's first wave lands on
// (scratch/uhwid.hip: of 1536 pairs of co-resident workgroups none had their waves 0 on one SIMD), so every chain wave
// issues alone. (Picking the wave by HW_ID slot instead put two of them on one SIMD: 88.3 us.)
// FULL: the block is exactly sixteen frame tiles (256 frames, what the pools of BASELINE cfg4 send): the instantiation carries
// the immediate-offset k-loop only. With both loops in one kernel their accumulators met in phi registers and every layer
// paid ~35 v_mov for it (found in the ISA; round 3) — an instruction count that matters here, see DESIGN.md §8.
template <bool FUSED, bool FULL>
__global__ __launch_bounds__(kConvmThreads, 4) void k_conv_mfma(LaunchArgs a, ConvDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int sg = blockIdx.x;
    const size_t rstride = a.row_stride ? a.row_stride : (size_t)n;      // a time slice of a longer block keeps the block's row pitch
    const int mode = FUSED ? (int)MODE_CHAIN : a.mode;
    if (!FUSED) {
        if (n == 0) return;
        if (mode == MODE_CHAIN) {
            const uint32_t flags = a.ctl[sg].flags;
            if (!(flags & CTL_ENABLED) || !(flags & CTL_NET_ON)) return;      // :607-619, :631-632 (uniform per workgroup)
            if (tid == 0) {
                StreamState& st = a.st[sg];
                st.pending = param_targets(a.ctl[sg], st, st.pending);
            }
        }
    }
    const int C = d.channels;
    const int Hb = (d.max_hist + 3) & ~3;                    // plane index of frame 0
    const int F = convm_plane_stride(d.max_hist, n);
    const size_t stage_floats = 2 * (size_t)d.max_k_steps * kWave;
    float* pl = smem;                                         // [C][F]: history | block, per channel
    float* wst = pl + (size_t)C * F;                          // [k_steps][64] records {B fragment value, A plane offset}
    float* wdl = wst + (stage_floats > (size_t)kChainHandFloats ? stage_floats : (size_t)kChainHandFloats);   // Dense weights [16] + bias
    float* verdict = wdl + 20;                                // fused form: {stream live, model in circuit, ChainCtx}
    constexpr int chain_wave = 0;

    const float* W = a.wpack;
    float* hist_base = a.nn + (size_t)sg * a.nn_stride;
    // Global-memory latency is kept off the layer loop: a layer's B fragments arrive in registers while the layer
    // before it computes, and so does the history prefix of the layer after it.
    constexpr int kFragRegs = 8;              // 32 k-steps x 64 lanes / 256 threads
    constexpr int kHistRegs = 8;              // 4096 history floats of a layer in registers; longer ones read directly
    // (the float4 and the scalar form of a history prefix in registers of their own: one set for both met in phi copies)
    float fr[kFragRegs], hp[kHistRegs], bias_r = 0.f;
    f32x4 hp4[kHistRegs / 4];
    int kcb[kFragRegs];                        // (cin << 16 | frames back) of the contraction row behind fragment element j
    auto fetch_frag = [&](int l) {
        const ConvLayer& L = d.L[l];
        const float2* recs = reinterpret_cast<const float2*>(W + (FULL ? L.wf_full_off : L.wf_off));
#pragma unroll
        for (int j = 0; j < kFragRegs; ++j) {
            if (j * kConvmThreads >= L.k_steps * kWave) break;
            const int i = tid + j * kConvmThreads;
            const float2 r = i < L.k_steps * kWave ? recs[i] : float2{ 0.f, 0.f };
            fr[j] = r.x;
            kcb[j] = __builtin_bit_cast(int, r.y);
        }
        bias_r = W[L.bs_off + (lane & 15)];
    };
    // what layer 0 needs besides its input row: its own short history in front of the row, the Dense weights
    auto layer0_side = [&](int t, int stride) {
        const ConvLayer& L0 = d.L[0];
        for (int c = 0; c < L0.in_ch; ++c)
            for (int j = t; j < L0.hist; j += stride)
                pl[c * F + convm_swz(c, Hb - L0.hist + j)] = hist_base[L0.state_off + c * L0.hist + j];
        if (t < 17) wdl[t] = t < 16 ? (t < d.L[d.n_layers - 1].out_ch ? W[d.wd_off + t] : 0.f) : W[d.bd_off];
    };

    bool net = true;
    if (wave == chain_wave) CV_STAMP(0);
#ifdef AIDAX_CONV_TRACE
    if (tid == 0) cv_trace()[13] = wall_clock64();
#endif
    if constexpr (FUSED) {
        // Before the pre pass, not after it: layer 0's fragments go into every thread's registers and the three waves that
        // have nothing to do until the barrier fetch what else layer 0 needs — reads only, whatever the prologue decides
        // (1.4 -> 0.x us between the pre pass and the first MFMA, scratch/conv_trace.py).
        fetch_frag(0);
        if (wave != chain_wave) layer0_side(tid - kWave, kConvmThreads - kWave);
        // channel 0's block part of the plane is the audio row: the pre pass leaves layer 0's input there
        if (wave == chain_wave) {
            ChainCtx ctx = chain_prologue<true>(a.ctl[sg], a.st[sg], a.in + (size_t)sg * rstride, a.out + (size_t)sg * rstride, pl + Hb, n, lane, wst);
            if (ctx.live && (ctx.flags & CTL_NET_ON)) {
                uint32_t pend = ctx.pending;
                if (lane == 0) pend = param_targets(a.ctl[sg], a.st[sg], pend);
                ctx.pending = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
            }
            if (lane == 0) {
                // the context waits in LDS, not in six registers of every thread across the layer loop
                verdict[0] = ctx.live ? 1.f : 0.f; verdict[1] = (ctx.live && (ctx.flags & CTL_NET_ON)) ? 1.f : 0.f;
                verdict[2] = __builtin_bit_cast(float, ctx.flags); verdict[3] = __builtin_bit_cast(float, ctx.pending);
                verdict[4] = ctx.pre_mem; verdict[5] = ctx.master_mem; verdict[6] = ctx.pre_tgt; verdict[7] = ctx.master_tgt;
            }
        }
        if (wave == chain_wave) CV_STAMP(4);
        lds_barrier();
        if (verdict[0] == 0.f) return;                        // pre-run / hard bypass: the prologue did all there is to do
        net = verdict[1] != 0.f;
    }

    float* row = mode == MODE_CHAIN ? a.out + (size_t)sg * rstride : a.out;
    const int n16 = (n + 15) & ~15;

    // layer-0 input: [history | x * in_gain | zeros up to the tile boundary]; x stays in a register for the skip
    float xg = 0.f;
    if (net) {
        if constexpr (!FUSED) layer0_side(tid, kConvmThreads);
        for (int t = tid; t < n16; t += kConvmThreads) {
            float v = 0.f;
            if (t < n) v = (FUSED ? pl[Hb + t] : mode == MODE_CHAIN ? row[t] : mode == MODE_NN_ONLY ? a.in[(size_t)t * a.input_size] : 0.f) * a.in_gain;
            pl[Hb + t] = v;
            xg = v;                                           // n <= 256: one frame per thread
        }
    }
    auto stage_frag = [&](int l) {            // registers -> one 8-byte record per (k-step, lane): B value, A plane offset
        const ConvLayer& L = d.L[l];
#pragma unroll
        for (int j = 0; j < kFragRegs; ++j) {
            if (j * kConvmThreads >= L.k_steps * kWave) break;
            const int i = tid + j * kConvmThreads;
            if (i < L.k_steps * kWave)      // the A offset in BYTES: the k-loop adds it to the plane's base as it is
                *reinterpret_cast<float2*>(wst + 2 * i) =
                    float2{ fr[j], FULL ? __builtin_bit_cast(float, kcb[j])      // (full blocks: the packer has done it)
                                        : __builtin_bit_cast(float, 4 * ((kcb[j] >> 16) * F + convm_swz(kcb[j] >> 16, Hb - (kcb[j] & 0xffff) + (i & 15)))) };
        }
    };
    // history of a layer: [in_ch][hist] in HBM <-> plane[ch][Hb-hist .. Hb), walked flat (coalesced, hist*in_ch/256
    // passes). i -> (ch, frame) without an integer division: (i + 0.5) / hist is at least 0.5/hist away from an
    // integer, far more than the fp32 error of the product, so the truncation is exact.
    // (`shift`: columns further right — save_history reads the last `hist` frames of [old history | block]; the swizzle of the
    // plane's columns, aidax_layout.h convm_swz, is applied to the final column)
    auto plane_index = [&](int i, int hist, float inv_hist, int shift) {
        const int ch = (int)(((float)i + 0.5f) * inv_hist);
        return ch * F + convm_swz(ch, Hb - hist + (i - ch * hist) + shift);
    };
    // A layer whose history length is a multiple of four frames moves it as float4s (16 B per lane, a quarter of the
    // index arithmetic): rows of `hist` floats per channel are contiguous on both sides and 16-byte aligned
    // (pack_conv aligns state_off, the plane geometry is a multiple of four).
    auto vec_hist = [&](const ConvLayer& L) { return (L.hist & 3) == 0 && (L.state_off & 3) == 0; };
    auto plane_index4 = [&](int i4, int hist4, float inv_hist4, int shift) {     // float4 index -> plane float offset of its first frame (shift: a multiple of 4)
        if ((hist4 & (hist4 - 1)) == 0) {                              // dilations are powers of two as a rule: shift and mask
            const int sh = 31 - __builtin_clz(hist4);
            return (i4 >> sh) * F + convm_swz(i4 >> sh, Hb - 4 * hist4 + 4 * (i4 & (hist4 - 1)) + shift);
        }
        const int ch = (int)(((float)i4 + 0.5f) * inv_hist4);
        return ch * F + convm_swz(ch, Hb - 4 * hist4 + 4 * (i4 - ch * hist4) + shift);
    };
    auto fetch_prefix = [&](int l) {
        const ConvLayer& L = d.L[l];
        const int cnt = L.hist * L.in_ch;
        if (vec_hist(L)) {
            const f32x4* src = reinterpret_cast<const f32x4*>(hist_base + L.state_off);
#pragma unroll
            for (int j = 0; j < kHistRegs / 4; ++j) {
                if (j * kConvmThreads * 4 >= cnt) break;
                const int i4 = tid + j * kConvmThreads;
                hp4[j] = 4 * i4 < cnt ? src[i4] : f32x4{ 0.f, 0.f, 0.f, 0.f };
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < kHistRegs; ++j) {
            if (j * kConvmThreads >= cnt) break;              // wave-uniform: short histories cost one pass, not sixteen
            const int i = tid + j * kConvmThreads;
            hp[j] = i < cnt ? hist_base[L.state_off + i] : 0.f;
        }
    };
    auto store_prefix = [&](int l) {
        const ConvLayer& L = d.L[l];
        const int cnt = L.hist * L.in_ch;
        const float inv = __builtin_amdgcn_rcpf((float)L.hist);      // (v_rcp: the index test below has a margin of 0.5 / hist, an IEEE divide is eleven instructions)
        if (vec_hist(L)) {
            const int hist4 = L.hist >> 2;
            const float inv4 = __builtin_amdgcn_rcpf((float)hist4);
#pragma unroll
            for (int j = 0; j < kHistRegs / 4; ++j) {
                if (j * kConvmThreads * 4 >= cnt) break;
                const int i4 = tid + j * kConvmThreads;
                if (4 * i4 < cnt)
                    *reinterpret_cast<f32x4*>(pl + plane_index4(i4, hist4, inv4, 0)) = hp4[j];
            }
            for (int i4 = tid + (kHistRegs / 4) * kConvmThreads; 4 * i4 < cnt; i4 += kConvmThreads)
                *reinterpret_cast<f32x4*>(pl + plane_index4(i4, hist4, inv4, 0)) = reinterpret_cast<const f32x4*>(hist_base + L.state_off)[i4];
            return;
        }
#pragma unroll
        for (int j = 0; j < kHistRegs; ++j) {
            if (j * kConvmThreads >= cnt) break;
            const int i = tid + j * kConvmThreads;
            if (i < cnt) pl[plane_index(i, L.hist, inv, 0)] = hp[j];
        }
        for (int i = tid + kHistRegs * kConvmThreads; i < cnt; i += kConvmThreads)
            pl[plane_index(i, L.hist, inv, 0)] = hist_base[L.state_off + i];
    };
    // this layer's new history: the last Hs frames of [old history | this block's inputs], plane -> HBM. Runs while the
    // plane still holds the layer's INPUT (before the barrier that releases it for the in-place update).
    auto save_history = [&](const ConvLayer& L) {
        const int Hs = L.hist, Ci = L.in_ch;
        if (vec_hist(L) && (n & 3) == 0) {
            const int hist4 = Hs >> 2, cnt4 = hist4 * Ci;
            const float inv4 = __builtin_amdgcn_rcpf((float)hist4);
            f32x4* dst = reinterpret_cast<f32x4*>(hist_base + L.state_off);
            f32x4 hv4[kHistRegs / 4];
#pragma unroll
            for (int j = 0; j < kHistRegs / 4; ++j) {         // all LDS reads first, then the stores
                if (j * kConvmThreads >= cnt4) break;
                const int i4 = tid + j * kConvmThreads;
                hv4[j] = *reinterpret_cast<const f32x4*>(pl + plane_index4(i4 < cnt4 ? i4 : 0, hist4, inv4, n));
            }
#pragma unroll
            for (int j = 0; j < kHistRegs / 4; ++j) {
                if (j * kConvmThreads >= cnt4) break;
                const int i4 = tid + j * kConvmThreads;
                if (i4 < cnt4) dst[i4] = hv4[j];
            }
            for (int i4 = tid + (kHistRegs / 4) * kConvmThreads; i4 < cnt4; i4 += kConvmThreads)
                dst[i4] = *reinterpret_cast<const f32x4*>(pl + plane_index4(i4, hist4, inv4, n));
        } else {
            const float inv = __builtin_amdgcn_rcpf((float)Hs);
            const int cnt = Hs * Ci;
            float hv[kHistRegs];
#pragma unroll
            for (int j = 0; j < kHistRegs; ++j) {
                if (j * kConvmThreads >= cnt) break;
                const int i = tid + j * kConvmThreads;
                hv[j] = pl[plane_index(i < cnt ? i : 0, Hs, inv, n)];
            }
#pragma unroll
            for (int j = 0; j < kHistRegs; ++j) {
                if (j * kConvmThreads >= cnt) break;
                const int i = tid + j * kConvmThreads;
                if (i < cnt) hist_base[L.state_off + i] = hv[j];
            }
            for (int i = tid + kHistRegs * kConvmThreads; i < cnt; i += kConvmThreads)
                hist_base[L.state_off + i] = pl[plane_index(i, Hs, inv, n)];
        }
    };

    // Issue priority by progress, FUSED form: a wave's priority steps down with every half layer (cyclically: there are four
    // levels), so that of the four workgroups that share a CU the ones that are behind go first. Left alone the SIMDs favour
    // their oldest waves: the workgroup dispatched last onto a CU came out of the layer loop 6.5 us after the first, and its
    // post pass — a lone wave — ended the launch with the rest of the CU idle (scratch/conv_trace.py; 67.5 -> 64.6 us for
    // BASELINE cfg4; a priority fixed per workgroup id is worse than none, 66.5 us). AIDAX_TUNE bit 2048 switches it off.
    auto set_prio = [&](int phase) {
        if (!FUSED || (AIDAX_TUNE(a) & 2048)) return;
        switch (phase & 3) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
        }
    };
    // the post pass's context (left in LDS by the prologue) and its coefficients and state: requested by the chain wave when
    // the last layer is done, used after the Dense
    ChainCtx post_ctx;
    ChainPass post_pass;
    auto post_begin = [&]() {
        post_ctx.live = true;
        post_ctx.flags = __builtin_bit_cast(uint32_t, verdict[2]); post_ctx.pending = __builtin_bit_cast(uint32_t, verdict[3]);
        post_ctx.pre_mem = verdict[4]; post_ctx.master_mem = verdict[5]; post_ctx.pre_tgt = verdict[6]; post_ctx.master_tgt = verdict[7];
        post_pass = chain_epilogue_begin(a.ctl[sg], a.st[sg], post_ctx, lane);
    };
    const int ntiles = n16 / 16;
    if (net) {
    if constexpr (!FUSED) fetch_frag(0);
    lds_barrier();                                            // layer 0's input is in the plane
    stage_frag(0);
    if (wave == chain_wave) CV_STAMP(5);
    for (int l = 0; l < d.n_layers; ++l) {
        const ConvLayer& L = d.L[l];
        const int Co = L.out_ch;
        const float bias = bias_r;                            // of layer l (fetch_frag below loads the next one's)
        set_prio(2 * l);
        if (l + 1 < d.n_layers) { fetch_frag(l + 1); fetch_prefix(l + 1); }
        lds_barrier();                                        // plane = this layer's input, wst = its fragments
        // tile j of this wave is frame tile wave + 4*j (n <= 256: at most kConvmTiles per wave). The accumulators START as the
        // result of the first k-step's MFMAs with the bias quad as their C operand — no sixteen moves to seed them.
        f32x4 acc[kConvmTiles];
        const f32x4 bias4 = f32x4{bias, bias, bias, bias};
        if (FULL || wave < ntiles) {
            const char* plw = reinterpret_cast<const char*>(pl + 16 * wave);
            if constexpr (FULL) {
                // a full block: the four tiles of a wave sit at fixed distances — one address per k-step, the rest
                // are the instruction's immediate offsets
                // (reading the next k-step's record ahead of this one's MFMAs was measured: 75.6 against 75.05 us)
                {
                    const float2 rec = *reinterpret_cast<const float2*>(wst + 2 * lane);
                    const float* ap = reinterpret_cast<const float*>(plw + __builtin_bit_cast(int, rec.y));
#pragma unroll
                    for (int j = 0; j < kConvmTiles; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[64 * j], rec.x, bias4, 0, 0, 0);
                }
                for (int kk = 1; kk < L.k_steps; ++kk) {
                    const float2 rec = *reinterpret_cast<const float2*>(wst + 2 * (kk * kWave + lane));
                    const float* ap = reinterpret_cast<const float*>(plw + __builtin_bit_cast(int, rec.y));
#pragma unroll
                    for (int j = 0; j < kConvmTiles; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[64 * j], rec.x, acc[j], 0, 0, 0);
                }
            } else {
                {
                    const float2 rec = *reinterpret_cast<const float2*>(wst + 2 * lane);
                    const float* ap = reinterpret_cast<const float*>(plw + __builtin_bit_cast(int, rec.y));
#pragma unroll
                    for (int j = 0; j < kConvmTiles; ++j) {
                        const float av = ap[wave + 4 * j < ntiles ? 64 * j : 0];   // tiles past the block re-read the first
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, rec.x, bias4, 0, 0, 0);
                    }
                }
                for (int kk = 1; kk < L.k_steps; ++kk) {
                    const float2 rec = *reinterpret_cast<const float2*>(wst + 2 * (kk * kWave + lane));
                    const float* ap = reinterpret_cast<const float*>(plw + __builtin_bit_cast(int, rec.y));
#pragma unroll
                    for (int j = 0; j < kConvmTiles; ++j) {
                        const float av = ap[wave + 4 * j < ntiles ? 64 * j : 0];
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, rec.x, acc[j], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kConvmTiles; ++j) acc[j] = bias4;
        }
        save_history(L);                                      // reads the input while it is still there
        lds_barrier();                                        // everybody is done reading the plane and wst
        set_prio(2 * l + 1);
        if (FULL || wave < ntiles) {
            const int co = lane & 15;
#pragma unroll
            for (int j = 0; j < kConvmTiles; ++j) {
                if ((!FULL && wave + 4 * j >= ntiles) || co >= Co) continue;
                f32x4 v = acc[j];
                if (L.activation == 1) { v.x = tanh_exp_pre(v.x); v.y = tanh_exp_pre(v.y); v.z = tanh_exp_pre(v.z); v.w = tanh_exp_pre(v.w); }
                else if (L.activation == 2) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
                else if (L.activation == 3) { v.x = fast_sigmoid(v.x); v.y = fast_sigmoid(v.y); v.z = fast_sigmoid(v.z); v.w = fast_sigmoid(v.w); }
                *reinterpret_cast<f32x4*>(pl + (size_t)co * F + convm_swz(co, Hb + 16 * (wave + 4 * j) + 4 * (lane >> 4))) = v;
            }
        }
        if (l + 1 < d.n_layers) { store_prefix(l + 1); stage_frag(l + 1); }
        if (wave == chain_wave && l == 0) CV_STAMP(6);
        if (wave == chain_wave && l == 3) CV_STAMP(7);
    }
    if (wave == chain_wave) CV_STAMP(8);
    lds_barrier();
    if constexpr (FUSED) { if (wave == chain_wave) post_begin(); }     // its loads travel while the Dense is computed
    // Dense(C,1) + skip / output gain (:171-181), one thread per frame. The fused form leaves the result where the
    // audio row was: a thread reads column `tid` of every channel and then overwrites column `tid` of channel 0.
    if (mode != MODE_WARMUP && tid < n) {
        const int Cl = d.L[d.n_layers - 1].out_ch;
        float y = wdl[16];
        for (int o = 0; o < Cl; ++o) y = __builtin_fmaf(wdl[o], pl[(size_t)o * F + convm_swz(o, Hb + tid)], y);
        float o2 = a.input_skip ? xg + y : y;
        if constexpr (FUSED) pl[Hb + tid] = o2 * a.out_gain;
        else row[tid] = o2 * a.out_gain;
    }
    }   // net
    else if constexpr (FUSED) { if (wave == chain_wave) post_begin(); }
    if constexpr (FUSED) {
        lds_barrier();                                        // the row is complete, the staging area is free again
        if (!(AIDAX_TUNE(a) & 2048)) __builtin_amdgcn_s_setprio(0);  // (the post pass: measured the same at priority 0 and 3)
        if (wave == chain_wave) CV_STAMP(9);
        if (wave == chain_wave) {
            chain_epilogue_run(a.st[sg], post_ctx, post_pass, a.out + (size_t)sg * rstride, pl + Hb, n, lane, wst);
#ifdef AIDAX_CONV_TRACE
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            CV_STAMP(12);
            __builtin_amdgcn_wave_barrier();
            if (lane < 13) (a.out + (size_t)sg * rstride)[lane] = __builtin_bit_cast(float, (uint32_t)(cv_trace()[lane] - cv_trace()[0]));
            if (lane == 13) (a.out + (size_t)sg * rstride)[13] = __builtin_bit_cast(float, (uint32_t)(cv_trace()[13] & 0xffffffffu));
            if (lane == 14) (a.out + (size_t)sg * rstride)[14] = __builtin_bit_cast(float, (uint32_t)(wall_clock64() & 0xffffffffu));
            if (lane == 15) (a.out + (size_t)sg * rstride)[15] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4));
            if (lane == 16) (a.out + (size_t)sg * rstride)[16] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20));
#endif
        }
    }
}

size_t convm_lds_bytes(const ConvDesc& d, uint32_t n_frames) { return convm_lds_floats(d, (int)n_frames) * sizeof(float); }

// Streams (= workgroups) of the fused form that are resident at once on `device`: while a pool fits, its chain passes
// cost their latency once; beyond, once per round of workgroups, and the packed k_chain launches are the cheaper form.
int convm_resident_streams(const ConvDesc& d, uint32_t n_frames, int device)
{
    int per_cu = 0, cus = 0;
    const size_t lds = convm_lds_bytes(d, n_frames);
    const void* fn = reinterpret_cast<const void*>(k_conv_mfma<true, false>);      // (both instantiations have the same footprint)
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kConvmThreads, lds) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu * cus;
}

// fused: the launch is the whole run() of every stream (a.mode must be MODE_CHAIN); otherwise applyModel only
hipError_t launch_conv_mfma_kernel(const LaunchArgs& a, const ConvDesc& d, bool fused, hipStream_t stream)
{
    if (fused && a.mode != MODE_CHAIN) return hipErrorInvalidValue;
    const size_t lds = convm_lds_bytes(d, a.n_frames);
    const bool full = a.n_frames == 16 * 4 * kConvmTiles;     // sixteen frame tiles: the instantiation without the ragged loop
    typedef void (*Fn)(LaunchArgs, ConvDesc);
    const Fn fn = fused ? (full ? k_conv_mfma<true, true> : k_conv_mfma<true, false>) : (full ? k_conv_mfma<false, true> : k_conv_mfma<false, false>);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3(a.n_streams), dim3(kConvmThreads), lds, stream, a, d);
    return hipGetLastError();
}

}  // namespace aidax
