// aidax_convs.hip — k_conv_ms: the causal dilated conv1d stack (BASELINE config #4) as bf16 TERM PRODUCTS on the matrix cores.
//
// k_conv_mfma (aidax_convm.hip) contracts a layer on v_mfma_f32_16x16x4_f32: 48 MFMAs of 32 cycles per layer and wave, and — what round 4
// measured (profiles/r04_overlap.txt) — the fp32 matrix instructions run at the vector rate and nothing else of the SIMD issues beside them,
// so a layer costs the SUM of its MFMA time and of its ~240 VALU instructions: 35 us of SIMD issue for BASELINE cfg4's 1024 streams.
// Here the same product runs the way k_gru_gs / k_mfma_ls run theirs: every fp32 operand split exactly into three bf16 terms
// (x = x0 + x1 + x2, 8 + 8 + 8 significant bits), six of the nine term products issued on v_mfma_f32_16x16x32_bf16 — the three dropped
// ones are together <= 2^-23 |w x|, below the rounding of the fp32 accumulation that follows — at 16 cycles per 8192 MACs and with the
// VALU free beside them. Per layer and wave: 48 bf16 MFMAs x 16 cycles instead of 48 fp32 x 32, the activations' VALU work in their shadow.
//
//   A (16 cout x 32 k)   the WEIGHTS: ready fragments from the packer (aidax_pack.cpp, ConvLayer::ms_*), split there; a k-step of 32 is
//                        two taps x sixteen input channels; 24 registers per layer, loaded while the layer before computes;
//   B (32 k x 16 frames) the ACTIVATIONS, straight out of the LDS plane with one ds_read_b128 per term: the plane is FRAME-major,
//                        [term][channel half][frame][8 bf16], so a lane's eight k values — eight channels of one tap at one frame —
//                        are sixteen contiguous bytes, and the sixteen lanes of a quarter wave read sixteen consecutive frames:
//                        256 contiguous bytes, every bank once;
//   D (16 cout x 16 frames) a lane holds four output channels of ONE frame: activation, split (gs_split4's arithmetic: two
//                        v_cvt_pk_bf16_f32 and four exact subtractions per term), one ds_write_b64 per term.
//
// ONE plane, updated in place, as in k_conv_mfma: a layer's sixteen frame tiles (four per wave) are accumulated in registers, every
// wave is done reading at a barrier, and only then the outputs overwrite the block part and the next layer's history the history part.
// The plane holds kConvsHist = 128 frames of history in front of the block's 256 (36 KiB: four workgroups per CU, all 1024 of cfg4
// resident at once); a layer that reaches further back (cfg4's last: dilation 128, 256 frames) takes those B fragments straight from
// its history in HBM — at most two (tile, k-step) pairs per wave, fetched into registers a layer ahead. The per-layer input history
// persists in HBM in the plane's own layout (96 bytes per frame), so a history is moved with 16-byte copies and read as fragments
// as it lies; it is NOT k_conv's / k_conv_mfma's layout — a pool decides for one family at model load (aidax_pool.cpp).
// Layer 0 (one scalar input, two to four taps) is a handful of fp32 FMAs per output; Dense(16,1) + skip / gain are fused into the
// last layer's epilogue (four FMAs and two permlane swaps on the lane's own fp32 activations).
// FUSED: the whole run() of the stream in the launch, the chain passes on wave 0 around the layers — k_conv_mfma's, line for line.
#include <type_traits>
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cs_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cs_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned cs_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCsThreads = 256;
constexpr int kCsTiles = 4;                          // frame tiles per wave (256 frames = 16 tiles = 4 waves x 4)
constexpr int kCsStrip = kConvsPF;                   // 16-byte vectors of one (term, channel half) strip of the plane
constexpr int kCsPlaneVecs = 6 * kCsStrip;
constexpr int kCsHandFloats = 2 * kChainWideLanes * kChainWideBlock;      // the chain wave's hand-over slots (chain_run_blocked)
constexpr int kCsMaxDeep = 2;                        // (tile, k-step) pairs of a wave that read beyond the plane (conv_ms_shape_ok)

__host__ __device__ constexpr size_t convs_lds_floats()
{
    return (size_t)kCsPlaneVecs * 4                  /* the plane; its LAST kCsHandFloats double as the chain passes' hand-over slots */
         + kConvsX0 + kConvsFrames                   /* audio row with layer 0's input history in front       */
         + 16 + 4                                    /* Dense weights + bias                                  */
         + 8;                                        /* fused form: what the chain prologue leaves for the epilogue */
}
static_assert(kCsHandFloats * 4 <= 64 * 16, "the hand-over slots sit in the last 64 frames of the last strip: written by layer 0's epilogue at the earliest");

__device__ __forceinline__ void cs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// a, b -> (bf16(a) | bf16(b) << 16), round to nearest even: ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cs_pack_bf16(float a, float b)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = { static_cast<__bf16>(a), static_cast<__bf16>(b) };
    unsigned u = __builtin_bit_cast(unsigned, v);
    asm volatile("" : "+v"(u));
    return u;
}
// four fp32 values -> three terms of four bf16 each, v = t0 + t1 + t2 exactly (k_gru_gs's gs_split4)
__device__ __forceinline__ void cs_split4(const f32x4& v, cs_u32x2 (&t)[3])
{
    float r[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned p01 = cs_pack_bf16(r[0], r[1]), p23 = cs_pack_bf16(r[2], r[3]);
        t[k] = cs_u32x2{ p01, p23 };
        if (k < 2) {
            r[0] -= __builtin_bit_cast(float, p01 << 16);
            r[1] -= __builtin_bit_cast(float, p01 & 0xffff0000u);
            r[2] -= __builtin_bit_cast(float, p23 << 16);
            r[3] -= __builtin_bit_cast(float, p23 & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ f32x4 cs_activate(f32x4 v, int activation)
{
    if (activation == 1) { v.x = tanh_exp_pre(v.x); v.y = tanh_exp_pre(v.y); v.z = tanh_exp_pre(v.z); v.w = tanh_exp_pre(v.w); }
    else if (activation == 2) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
    else if (activation == 3) { v.x = fast_sigmoid(v.x); v.y = fast_sigmoid(v.y); v.z = fast_sigmoid(v.z); v.w = fast_sigmoid(v.w); }
    return v;
}

// FULL: the block is exactly sixteen frame tiles (256 frames, what the pools of BASELINE cfg4 send): no tile is ragged or absent, and the
// k-loop is straight-line code.
template <bool FUSED, bool FULL>
__global__ __launch_bounds__(kCsThreads, 4) void k_conv_ms(LaunchArgs a, ConvDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, nl = lane & 15;
    const int n = FULL ? kConvsFrames : (int)a.n_frames;
    const int sg = blockIdx.x;
    const size_t rstride = a.row_stride ? a.row_stride : (size_t)n;
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
    constexpr int Hh = kConvsHist;
    const int n16 = FULL ? kConvsFrames : (n + 15) & ~15;
    const int ntiles = FULL ? 16 : n16 / 16;
    cs_u32x4* pl = reinterpret_cast<cs_u32x4*>(smem);         // [term * 2 + half][kCsStrip]: frame f at index f + Hh
    float* xbuf = smem + (size_t)kCsPlaneVecs * 4;            // [kConvsX0 frames of layer 0's input history | the audio row]
    float* xrow = xbuf + kConvsX0;
    float* wdl = xrow + kConvsFrames;                         // Dense weights [16] + bias
    float* verdict = wdl + 20;
    float* hand = smem + (size_t)kCsPlaneVecs * 4 - kCsHandFloats;
    constexpr int chain_wave = 0;
    const int NL = d.n_layers;

    const float* W = a.wpack;
    float* st_base = a.nn + (size_t)sg * a.nn_stride;
    const ConvLayer& L0 = d.L[0];

    // ---- register prefetches: a layer's A fragments, the in-plane part of its input history, its deep B fragments
    cs_u32x4 afr[2][3];                                       // [k-step][term]
    auto fetch_afrags = [&](int l) {
        const ConvLayer& L = d.L[l];
        const cs_u32x4* rec = reinterpret_cast<const cs_u32x4*>(W + L.ms_w_off) + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                afr[ks][t] = ks < L.ms_ksteps ? rec[(ks * 3 + t) * kWave] : cs_u32x4{ 0u, 0u, 0u, 0u };
    };
    // in-plane history of layer l: the last hp = min(hist, Hh) frames of its six strips, 6 hp <= 768 vectors, three per thread
    cs_u32x4 hpv[3];
    auto prefix_index = [&](int v, int hp, float inv_hp, int& strip, int& k) {      // v -> (strip, frame k of the hp): (v + 0.5) / hp is >= 0.5 / hp off an integer
        strip = (int)(((float)v + 0.5f) * inv_hp);
        k = v - strip * hp;
    };
    auto fetch_prefix = [&](int l) {
        const ConvLayer& L = d.L[l];
        const int hp = L.hist < Hh ? L.hist : Hh;
        const float inv_hp = __builtin_amdgcn_rcpf((float)hp);
        const cs_u32x4* src = reinterpret_cast<const cs_u32x4*>(st_base + L.ms_state_off);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int v = tid + r * kCsThreads;
            int strip, k;
            prefix_index(v < 6 * hp ? v : 0, hp, inv_hp, strip, k);
            hpv[r] = v < 6 * hp ? src[strip * L.hist + (L.hist - hp + k)] : cs_u32x4{ 0u, 0u, 0u, 0u };
        }
    };
    auto store_prefix = [&](int l) {
        const ConvLayer& L = d.L[l];
        const int hp = L.hist < Hh ? L.hist : Hh;
        const float inv_hp = __builtin_amdgcn_rcpf((float)hp);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int v = tid + r * kCsThreads;
            if (v < 6 * hp) {
                int strip, k;
                prefix_index(v, hp, inv_hp, strip, k);
                pl[strip * kCsStrip + (Hh - hp + k)] = hpv[r];
            }
        }
    };
    // this lane's source in k-step ks of layer L: frames back and channel half of its quarter's tap (padding quarters read their own frame)
    // (both halves' values are read uniformly and selected per lane: no per-lane index into the descriptor)
    auto lane_raw = [&](const ConvLayer& L, int ks) { const int s0 = L.ms_shift[ks][0], s1 = L.ms_shift[ks][1]; return (q >> 1) ? s1 : s0; };
    auto lane_shift = [&](const ConvLayer& L, int ks) { const int s = lane_raw(L, ks); return s < 0 ? 0 : s; };
    auto lane_pad = [&](const ConvLayer& L, int ks) { return lane_raw(L, ks) < 0; };
    // does tile t of k-step ks reach beyond the plane's history? (wave-uniform: from the descriptor)
    auto pair_deep = [&](const ConvLayer& L, int ks, int t) {
        const int s0 = L.ms_shift[ks][0], s1 = L.ms_shift[ks][1];
        return (s0 >= 0 && 16 * t - s0 < -Hh) || (s1 >= 0 && 16 * t - s1 < -Hh);
    };
    // Deep B fragments of the next layer to run: only the FIRST k-step (the oldest taps) of a wave's first two tiles can reach beyond the
    // plane (conv_ms_shape_ok admits no other stack), so the two register sets have fixed owners — (k-step 0, tile j = 0) and (0, j = 1).
    cs_u32x4 dfr0[3], dfr1[3];
    auto fetch_deep = [&](int l) {
        const ConvLayer& L = d.L[l];
        if (L.hist <= Hh) return;
        const cs_u32x4* src = reinterpret_cast<const cs_u32x4*>(st_base + L.ms_state_off) + (q & 1) * L.hist;
        const int sh = lane_shift(L, 0);
        const bool pad = lane_pad(L, 0);
#pragma unroll
        for (int j = 0; j < kCsMaxDeep; ++j) {
            const int t = wave + 4 * j;
            if (t >= ntiles || !pair_deep(L, 0, t)) continue;
            const int f = 16 * t + nl - sh;                                // this lane's source frame
            const int hi = (f < -Hh && !pad) ? L.hist + f : 0;             // frame f of the history = index hist + f
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                const cs_u32x4 v = src[(size_t)term * 2 * L.hist + hi];
                if (j == 0) dfr0[term] = v; else dfr1[term] = v;
            }
        }
    };

    // Everything layer 0 and layer 1 need from memory is requested HERE, before the pre pass: layer 0's weights and bias (a lane's four
    // channels: tanh layers carry 2 log2 e, like the records), layer 1's history prefix, A fragments and deep fragments — reads only,
    // whatever the prologue decides. (Requested where they are used, layer 0 — a few FMAs per output — took as long as a layer on the
    // matrix cores: scratch/conv_trace.py.)
    f32x4 w0r[4], b0;
    auto fetch_layer0 = [&]() {
        const float scale = L0.activation == 1 ? kTwoLog2e : 1.0f;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const float* wp = W + L0.w_off + (size_t)(tap < L0.ksize ? tap : 0) * 16 + 4 * q;
            const float m = tap < L0.ksize ? scale : 0.f;
            w0r[tap] = f32x4{ m * wp[0], m * wp[1], m * wp[2], m * wp[3] };
        }
        const float* bp = W + L0.bs_off + 4 * q;
        b0 = f32x4{ bp[0], bp[1], bp[2], bp[3] };
    };
    // (the chain wave has no registers to spare across the pre pass, nor has the ragged-block instantiation: they ask behind it)
    constexpr bool kEarly = FULL;
    const bool early_wave = kEarly && (!FUSED || wave != chain_wave);
    if (early_wave) { fetch_layer0(); if (NL > 1) { fetch_prefix(1); fetch_afrags(1); } }
    bool net = true;
    if (wave == chain_wave) CV_STAMP(0);
#ifdef AIDAX_CONV_TRACE
    if (tid == 0) cv_trace()[13] = wall_clock64();
#endif
    if constexpr (FUSED) {
        if (wave != chain_wave) {
            // what the layers need besides the audio row, fetched while the chain wave runs the pre pass: layer 0's input history,
            // the Dense weights (reads only, whatever the prologue decides)
            const int t = tid - kWave;
            for (int j = t; j < L0.hist; j += kCsThreads - kWave) xbuf[kConvsX0 - L0.hist + j] = st_base[L0.ms_state_off + j];
            if (t < 17) wdl[t] = t < 16 ? W[d.wd_off + t] : W[d.bd_off];
        }
        if (wave == chain_wave) {
            ChainCtx ctx = chain_prologue<true>(a.ctl[sg], a.st[sg], a.in + (size_t)sg * rstride, a.out + (size_t)sg * rstride, xrow, n, lane, hand);
            if (ctx.live && (ctx.flags & CTL_NET_ON)) {
                uint32_t pend = ctx.pending;
                if (lane == 0) pend = param_targets(a.ctl[sg], a.st[sg], pend);
                ctx.pending = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
            }
            if (lane == 0) {
                verdict[0] = ctx.live ? 1.f : 0.f; verdict[1] = (ctx.live && (ctx.flags & CTL_NET_ON)) ? 1.f : 0.f;
                verdict[2] = __builtin_bit_cast(float, ctx.flags); verdict[3] = __builtin_bit_cast(float, ctx.pending);
                verdict[4] = ctx.pre_mem; verdict[5] = ctx.master_mem; verdict[6] = ctx.pre_tgt; verdict[7] = ctx.master_tgt;
            }
        }
        if (wave == chain_wave) CV_STAMP(4);
        cs_lds_barrier();
        if (verdict[0] == 0.f) return;                        // pre-run / hard bypass: the prologue did all there is to do
        net = verdict[1] != 0.f;
    }

    float* row = mode == MODE_CHAIN ? a.out + (size_t)sg * rstride : a.out;
    ChainCtx post_ctx;
    ChainPass post_pass;
    auto post_begin = [&]() {
        post_ctx.live = true;
        post_ctx.flags = __builtin_bit_cast(uint32_t, verdict[2]); post_ctx.pending = __builtin_bit_cast(uint32_t, verdict[3]);
        post_ctx.pre_mem = verdict[4]; post_ctx.master_mem = verdict[5]; post_ctx.pre_tgt = verdict[6]; post_ctx.master_tgt = verdict[7];
        post_pass = chain_epilogue_begin(a.ctl[sg], a.st[sg], post_ctx, lane);
    };

    if (net) {
        if constexpr (!FUSED) {
            for (int j = tid; j < L0.hist; j += kCsThreads) xbuf[kConvsX0 - L0.hist + j] = st_base[L0.ms_state_off + j];
            if (tid < 17) wdl[tid] = tid < 16 ? W[d.wd_off + tid] : W[d.bd_off];
        }
        // layer-0 input: x * in_gain, zeros up to the tile boundary
        for (int t = tid; t < n16; t += kCsThreads) {
            float v = 0.f;
            if (t < n) v = (FUSED ? xrow[t] : mode == MODE_CHAIN ? row[t] : mode == MODE_NN_ONLY ? a.in[(size_t)t * a.input_size] : 0.f) * a.in_gain;
            xrow[t] = v;
        }
        if (!early_wave) { fetch_layer0(); if (NL > 1) { fetch_prefix(1); fetch_afrags(1); } }
        cs_lds_barrier();                                     // the row and layer 0's history are in xbuf
        if (wave == chain_wave) CV_STAMP(5);

        // a lane's outputs: channels 4q .. 4q+3 of frame 16 t + nl -> its plane slot (three 8-byte term writes) / the Dense
        const float* wdq = wdl + 4 * q;
        auto emit = [&](bool last_layer, int t, f32x4 v) {
            int f = 16 * t + nl;
            asm volatile("" : "+v"(f));                       // (opaque: otherwise the four tiles' row addresses are hoisted out of the layer loop and live — in scratch — across it)
            if (!last_layer) {
                if (f >= n) return;
                cs_u32x2 tm[3];
                cs_split4(v, tm);
                cs_u32x2* dst = reinterpret_cast<cs_u32x2*>(pl + (q >> 1) * kCsStrip + (f + Hh)) + (q & 1);
#pragma unroll
                for (int term = 0; term < 3; ++term) dst[(size_t)term * 2 * kCsStrip * 2] = tm[term];
            } else if (mode != MODE_WARMUP) {
                // Dense(16,1) + skip / output gain (:171-181): the lane's four channels, then the four quarters of the wave
                float y = wdq[0] * v.x;
                y = __builtin_fmaf(wdq[1], v.y, y);
                y = __builtin_fmaf(wdq[2], v.z, y);
                y = __builtin_fmaf(wdq[3], v.w, y);
                const Pair r2 = share_rows(y);
                y = r2.lo + r2.hi;
                const Pair r4 = share_halves(y);
                y = (r4.lo + r4.hi) + wdl[16];
                // the frame's model input sits in the row still (read before the row is overwritten: every lane of the frame reads, lane q == 0 writes)
                const float x = xrow[f < n16 ? f : 0];
                const float o = (a.input_skip ? x + y : y) * a.out_gain;
                __builtin_amdgcn_wave_barrier();
                if (q == 0 && f < n) {
                    if constexpr (FUSED) xrow[f] = o;
                    else row[f] = o;
                }
            }
        };

        // ---- layer 0: one scalar input, ksize <= 4 taps -> sixteen channels: fp32 FMAs (tanh layers carry 2 log2 e, like the records)
        {
            // layer 0's new history: the last hist frames of [old history | this block's input] (xbuf is not written before the Dense)
            for (int i = tid; i < L0.hist; i += kCsThreads) st_base[L0.ms_state_off + i] = xbuf[kConvsX0 + n - L0.hist + i];
#pragma unroll
            for (int j = 0; j < kCsTiles; ++j) {
                const int t = wave + 4 * j;
                if (t >= ntiles) continue;
                f32x4 v = b0;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    if (tap >= L0.ksize) break;
                    const float x = xrow[16 * t + nl - (L0.ksize - 1 - tap) * L0.dilation];
                    v.x = __builtin_fmaf(w0r[tap].x, x, v.x); v.y = __builtin_fmaf(w0r[tap].y, x, v.y);
                    v.z = __builtin_fmaf(w0r[tap].z, x, v.z); v.w = __builtin_fmaf(w0r[tap].w, x, v.w);
                }
                emit(NL == 1, t, cs_activate(v, L0.activation));
            }
            if (NL > 1) { store_prefix(1); fetch_deep(1); }   // (a second layer that reaches beyond the plane would be an odd stack; its fragments are asked for late)
            if (wave == chain_wave) CV_STAMP(6);
        }

        // ---- layers 1 .. NL-1 on the matrix cores
        // (the last layer's copy of the body is its own: what it alone does — the Dense, the post pass's loads — would otherwise be
        // carried through every iteration: the post pass's coefficients and state, ~28 registers, undefined until the last one)
        // Issue priority by progress (k_conv_mfma's, for the same reason: left alone a SIMD favours its oldest waves, the workgroup
        // dispatched last onto a CU leaves the layer loop microseconds behind the first, and its post pass — a lone wave — ends the
        // launch with the rest of the CU idle): a wave's priority steps down with every half layer, cyclically over the four levels, so
        // that of the four workgroups of a CU the ones that are behind go first. AIDAX_TUNE bit 2048 switches it off.
        auto set_prio = [&](int phase) {
            if (!FUSED || (AIDAX_TUNE(a) & 2048)) return;
            switch (phase & 3) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
        };
        auto layer = [&](int l, auto last_tag) {
            constexpr bool kLast = decltype(last_tag)::value;
            const ConvLayer& L = d.L[l];
            set_prio(2 * l);
            const bool deep_layer = L.hist > Hh;
            cs_lds_barrier();                                 // the plane holds this layer's input, history part included
            if (wave == chain_wave && l == 4) CV_STAMP(16);
            // This layer's new history — the last hist frames of [old history | this block's input] — leaves FIRST, while the input is in the
            // plane: six 16-byte reads per thread (one per strip; two for histories beyond 256 frames), then the stores, which nobody
            // waits for. (Behind the k-loop, strip by strip, this took as long as the k-loop itself: six LDS -> HBM round trips in a row
            // with every SIMD idle — scratch/conv_trace.py.) A deep layer's fragments came out of the history this overwrites, requested a
            // layer ago by every wave for itself: they must have arrived everywhere first.
            {
                cs_u32x4* hbm = reinterpret_cast<cs_u32x4*>(st_base + L.ms_state_off);
                if (deep_layer) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cs_lds_barrier(); }
                const bool from_old = L.hist > Hh + n;        // a short block on a deep layer: part of the new history is old history, moved up
                if (!from_old) {
                    cs_u32x4 hv[6];
                    const int i = tid < L.hist ? tid : 0;
                    const int f = n - L.hist + i;
#pragma unroll
                    for (int strip = 0; strip < 6; ++strip) hv[strip] = pl[strip * kCsStrip + f + Hh];
                    if (tid < L.hist) {
#pragma unroll
                        for (int strip = 0; strip < 6; ++strip) hbm[strip * L.hist + i] = hv[strip];
                    }
                    for (int i2 = tid + kCsThreads; i2 < L.hist; i2 += kCsThreads)      // histories of more than 256 frames
                        for (int strip = 0; strip < 6; ++strip) hbm[strip * L.hist + i2] = pl[strip * kCsStrip + n - L.hist + i2 + Hh];
                } else {
                    for (int strip = 0; strip < 6; ++strip) {
                        cs_u32x4 hv[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int i = tid + r * kCsThreads;
                            if (r * kCsThreads >= L.hist) break;
                            const int f = n - L.hist + (i < L.hist ? i : 0);
                            hv[r] = pl[strip * kCsStrip + (f >= -Hh ? f + Hh : 0)];
                            if (f < -Hh) hv[r] = hbm[strip * L.hist + (i < L.hist ? i + n : 0)];      // (separate statements: an LDS and a global address in one select make a flat pointer)
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        cs_lds_barrier();                     // everybody has read this strip's old frames
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int i = tid + r * kCsThreads;
                            if (r * kCsThreads >= L.hist) break;
                            if (i < L.hist) hbm[strip * L.hist + i] = hv[r];
                        }
                    }
                }
            }
            f32x4 acc[kCsTiles];
            {
                const float* bp = W + L.bs_off + 4 * q;
                const f32x4 b4 = f32x4{ bp[0], bp[1], bp[2], bp[3] };
#pragma unroll
                for (int j = 0; j < kCsTiles; ++j) acc[j] = b4;
            }
            // The k-loop over the wave's (k-step, tile) pairs p = 4 ks + j, software-pipelined by hand: a pair's three fragment reads are
            // requested before the pair in front of it multiplies, so that no MFMA group waits for its ds_read_b128s (one pair at a time,
            // reads then MFMAs, a wave on its own spent more time waiting for LDS than multiplying: scratch/conv_trace.py). Tiles past a
            // ragged block's end are multiplied all the same — on whatever the plane holds there — and never stored: no branch in here.
            const int sh0 = lane_shift(L, 0), sh1 = lane_shift(L, 1);
            const bool pad0 = lane_pad(L, 0);
            const cs_u32x4* plq = pl + (q & 1) * kCsStrip + nl + Hh;
            auto load_b = [&](int p, cs_u32x4 (&b)[3]) {
                const int idx = 16 * (wave + 4 * (p & 3)) - (p < 4 ? sh0 : sh1);      // this lane's source frame, less nl
                const cs_u32x4* src = plq + (idx + nl + Hh > 0 ? idx : -(nl + Hh));
#pragma unroll
                for (int term = 0; term < 3; ++term) b[term] = src[(size_t)term * 2 * kCsStrip];
            };
            auto mul_b = [&](int p, cs_u32x4 (&b)[3]) {
                const int j = p & 3, ks = p >> 2;
                if (deep_layer && p < kCsMaxDeep && pair_deep(L, 0, wave + 4 * j)) {
                    const bool mine = 16 * (wave + 4 * j) + nl - sh0 < -Hh && !pad0;
#pragma unroll
                    for (int term = 0; term < 3; ++term) {
                        const cs_u32x4 dv = j == 0 ? dfr0[term] : dfr1[term];
                        if (mine) b[term] = dv;
                    }
                }
                // (w0 w1 w2) x0 | (w0 w1) x1 | w0 x2: the six term products, the large ones first
#pragma unroll
                for (int th = 0; th < 3; ++th)
#pragma unroll
                    for (int tw = 0; tw < 3 - th; ++tw)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cs_bf16x8, afr[ks][tw]), __builtin_bit_cast(cs_bf16x8, b[th]), acc[j], 0, 0, 0);
            };
            {
                const int NP = 4 * L.ms_ksteps;               // 4 or 8 pairs
                cs_u32x4 bA[3], bB[3];
                load_b(0, bA);
#pragma unroll
                for (int p = 0; p < 8; p += 2) {
                    if (p >= NP) break;
                    load_b(p + 1, bB);
                    __builtin_amdgcn_sched_barrier(0);
                    mul_b(p, bA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 2 < NP) load_b(p + 2, bA);
                    __builtin_amdgcn_sched_barrier(0);
                    mul_b(p + 1, bB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (wave == chain_wave && l == 4) CV_STAMP(17);
            // the next layer's A fragments (L2-resident, shared by every stream) and deep B fragments: requested now, used a barrier,
            // an epilogue and another barrier later
            if constexpr (!kLast) { fetch_prefix(l + 1); fetch_afrags(l + 1); fetch_deep(l + 1); }      // (the history prefix first: it comes from HBM)
            if (wave == chain_wave && l == 4) CV_STAMP(18);
            cs_lds_barrier();                                 // everybody is done reading the plane
            set_prio(2 * l + 1);
            if (wave == chain_wave && l == 4) CV_STAMP(19);
            if constexpr (FUSED && kLast) { if (wave == chain_wave) post_begin(); }     // its loads travel while the last layer's outputs are made
#pragma unroll
            for (int j = 0; j < kCsTiles; ++j) {
                const int t = wave + 4 * j;
                if (t < ntiles) emit(kLast, t, cs_activate(acc[j], L.activation));
            }
            if (wave == chain_wave && l == 4) CV_STAMP(20);
            if constexpr (!kLast) store_prefix(l + 1);
            if (wave == chain_wave && l == 4) CV_STAMP(21);
            if (wave == chain_wave && l == 3) CV_STAMP(7);
            if (wave == chain_wave && kLast) CV_STAMP(8);
        };
        // (Measured and not kept: two of a CU's four workgroups started half a layer late, with the priority cycle shifted to match, so
        // that one pair's k-loop would meet the other pair's epilogue — 55.6 against 55.5 us. A bf16 MFMA hides two VALU instructions,
        // not the sixteen cycles' worth: an epilogue's ~220 barely fit under the other pair's 96 MFMAs, and a layer costs close to the sum.)
        for (int l = 1; l + 1 < NL; ++l) layer(l, std::false_type{});
        if (NL > 1) layer(NL - 1, std::true_type{});
        if (NL == 1) {                                        // (conv_ms_shape_ok asks for two layers; kept for completeness)
            if constexpr (FUSED) { if (wave == chain_wave) post_begin(); }
        }
    }   // net
    else if constexpr (FUSED) { if (wave == chain_wave) post_begin(); }
    if constexpr (FUSED) {
        cs_lds_barrier();                                     // the row is complete, the plane (and with it the hand-over slots) is free
        if (!(AIDAX_TUNE(a) & 2048)) __builtin_amdgcn_s_setprio(0);
        if (wave == chain_wave) CV_STAMP(9);
        if (wave == chain_wave) {
            chain_epilogue_run(a.st[sg], post_ctx, post_pass, a.out + (size_t)sg * rstride, xrow, n, lane, hand);
#ifdef AIDAX_CONV_TRACE
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            CV_STAMP(12);
            __builtin_amdgcn_wave_barrier();
            float* orow = a.out + (size_t)sg * rstride;
            if (lane < 13 || (lane >= 16 && lane < 22)) orow[lane] = __builtin_bit_cast(float, (uint32_t)(cv_trace()[lane] - cv_trace()[0]));
            if (lane == 13) orow[13] = __builtin_bit_cast(float, (uint32_t)(cv_trace()[13] & 0xffffffffu));
            if (lane == 14) orow[14] = __builtin_bit_cast(float, (uint32_t)(wall_clock64() & 0xffffffffu));
#endif
        }
    }
}

size_t convs_lds_bytes() { return convs_lds_floats() * sizeof(float); }

int convs_resident_streams(int device)
{
    int per_cu = 0, cus = 0;
    const void* fn = reinterpret_cast<const void*>(k_conv_ms<true, false>);      // (both instantiations have the same footprint)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kCsThreads, convs_lds_bytes()) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu * cus;
}

// fused: the launch is the whole run() of every stream (a.mode must be MODE_CHAIN); otherwise applyModel only. n_frames <= 256.
hipError_t launch_conv_ms_kernel(const LaunchArgs& a, const ConvDesc& d, bool fused, hipStream_t stream)
{
    if (!d.ms_ok || a.n_frames > (uint32_t)kConvsFrames || (fused && a.mode != MODE_CHAIN)) return hipErrorInvalidValue;
    const bool full = a.n_frames == (uint32_t)kConvsFrames;
    typedef void (*Fn)(LaunchArgs, ConvDesc);
    const Fn fn = fused ? (full ? k_conv_ms<true, true> : k_conv_ms<true, false>) : (full ? k_conv_ms<false, true> : k_conv_ms<false, false>);
    hipLaunchKernelGGL(fn, dim3(a.n_streams), dim3(kCsThreads), convs_lds_bytes(), stream, a, d);
    return hipGetLastError();
}

}  // namespace aidax
