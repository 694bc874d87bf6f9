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
    // A layer's history in HBM is a RING (round 6): the frame at absolute time tau of the stream sits at index tau mod hist of every strip, `pos` = the
    // time of this block's first frame modulo the stack's common period (ConvDesc::ms_pos_mod). A block then costs its own frames and nothing else —
    // no history is moved up when a block is shorter than it — and k_conv_st can take blocks of any tile count on the same state.
    uint32_t* pos_slot = reinterpret_cast<uint32_t*>(st_base + d.ms_pos_off);
    const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)*pos_slot);
    auto head_of = [&](const ConvLayer& L) { return (int)(pos % (uint32_t)L.hist); };       // where frame 0 of this block goes

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
        const int head = head_of(L);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int v = tid + r * kCsThreads;
            int strip, k;
            prefix_index(v < 6 * hp ? v : 0, hp, inv_hp, strip, k);
            int i = head - hp + k;                                  // frame k - hp of the block's time
            i = i < 0 ? i + L.hist : i;
            hpv[r] = v < 6 * hp ? src[strip * L.hist + i] : cs_u32x4{ 0u, 0u, 0u, 0u };
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
    // Where a lane's B fragment number i of k-step ks sits, in units of TS = the distance between two terms: term i — except in a PACKED first
    // k-step (three taps: the lone oldest tap's six term products in three instructions, ConvLayer::ms_packed0), where instruction i takes
    // term {0, 0, 1}[i] in the k-step's first half and {0, 1, 2}[i] in its second.
    auto b_term_off = [&](const ConvLayer& L, int ks, int i, int TS) {
        if (ks != 0 || !L.ms_packed0) return i * TS;
        return i == 0 ? 0 : (i - 1) * TS + (q >= 2 ? TS : 0);
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
        const int head = head_of(L);
#pragma unroll
        for (int j = 0; j < kCsMaxDeep; ++j) {
            const int t = wave + 4 * j;
            if (t >= ntiles || !pair_deep(L, 0, t)) continue;
            const int f = 16 * t + nl - sh;                                // this lane's source frame
            int hi = (f < -Hh && !pad) ? head + f : 0;                     // frame f < 0 of the history: index (head + f) mod hist
            hi = hi < 0 ? hi + L.hist : hi;
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                const cs_u32x4 v = src[b_term_off(L, 0, term, 2 * L.hist) + hi];
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
                // the ring takes the block's last c = min(n, hist) frames, frame n - c + i at index (start + i) mod hist: a block shorter than the
                // history leaves the rest of it where it is
                const int c = n < L.hist ? n : L.hist;
                const int start = (int)((pos + (uint32_t)(n - c)) % (uint32_t)L.hist);
                cs_u32x4 hv[6];
                const int i = tid < c ? tid : 0;
#pragma unroll
                for (int strip = 0; strip < 6; ++strip) hv[strip] = pl[strip * kCsStrip + (n - c + i) + Hh];
                if (tid < c) {
                    int j = start + i;
                    j = j >= L.hist ? j - L.hist : j;
#pragma unroll
                    for (int strip = 0; strip < 6; ++strip) hbm[strip * L.hist + j] = hv[strip];
                }
                for (int i2 = tid + kCsThreads; i2 < c; i2 += kCsThreads) {      // histories of more than 256 frames
                    int j = start + i2;
                    j = j >= L.hist ? j - L.hist : j;
                    for (int strip = 0; strip < 6; ++strip) hbm[strip * L.hist + j] = pl[strip * kCsStrip + (n - c + i2) + Hh];
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
                for (int term = 0; term < 3; ++term) b[term] = src[b_term_off(L, p >> 2, term, 2 * kCsStrip)];
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
                if (ks == 0 && L.ms_packed0) {
                    // the packed first k-step: [w0 | w1] x [x0 | x0], [w2 | w0] x [x0 | x1], [w1 | w0] x [x1 | x2] — fragments as the packer and load_b laid them
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cs_bf16x8, afr[0][i]), __builtin_bit_cast(cs_bf16x8, b[i]), acc[j], 0, 0, 0);
                    return;
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
        if (tid == 0) *pos_slot = (pos + (uint32_t)n) % d.ms_pos_mod;      // the stream's time moves on by the block (every layer's ring with it)
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

// ------------------------------------------------------------------------------------------------------------------------------------
// k_conv_st: the same run() of the same stack as a STREAM — tile-major instead of layer-major. k_conv_ms's workgroup starts with the
// pre pass (a lone wave, 8 us), ends with the post pass (10 us) and shares its CU's SIMDs with three other workgroups doing the same at
// the same moments: 18 of a workgroup's 47 us are one wave per SIMD issuing dependent fp64 instructions with nothing beside it. Here the
// block moves through the stack sixteen frames (one MFMA column tile) at a time, a TICK per tile and pipeline stage, one barrier per
// tick:
//     wave 0   both chain passes AT ONCE: the pre pass on lanes 0 .. 5, the post pass on lanes 6 .. 11 of the same instructions
//              (chain_macro_step: a stage per lane, eight frames per hand-over, two steps per tick) — tile j of the input at tick j, tile j
//              of the output d1 + 3 ticks later;
//     wave 1   layer 0 (fp32 FMAs), layers 1, 2         wave 2   layers 3, 4, 5         wave 3   layers 6, 7, Dense + skip / gain,
// each a tick behind the one in front of it (FOUR waves: a fifth would put two of a workgroup's waves on one SIMD, and the CU does not then
// take four workgroups — measured: 768 of cfg4's 1024 resident), a wave's layers back to back on the same tile (LDS runs a wave's accesses in order). A
// layer's input lives in a RING of its own in LDS — its history plus the tiles in flight, in k_conv_ms's [term][channel half][frame]
// vectors — so nothing is copied between layers, no history goes out and in through HBM per layer pass, and a wave keeps the A fragments
// of its layers in registers for the whole launch. 22 tiles of rings = 33 KiB: four workgroups per CU as before (40 928 of a quarter CU's
// 40 960 bytes with the row, the small tables and the staging area). What no ring holds comes from HBM as B fragments WITHOUT passing through
// registers (global_load_lds_dwordx4 into a 4.5 KiB staging area, requested when the tile before has been read): layer 6's oldest tap (128
// frames back) and both old taps of layer 7 (256, 128) — out of the layer's history where the frame belongs to the block before (k_conv_ms's
// layout: the two kernels share the state, ragged blocks and the split form go through k_conv_ms), out of what this launch has written
// where it belongs to this one (layer 7's history IS the block; layer 6's takes the block's second half, the first goes to 12 KiB of
// scratch per stream). The new histories leave as the tiles pass: the wave that reads a tile's newest tap has the very vectors the history
// consists of in registers. Same fragments, same MFMA order, same epilogue arithmetic as k_conv_ms: the outputs are bit-identical to it.
// How it got here, step by step with every measurement: profiles/r05_cfg4_stream_steps.txt; its timeline: profiles/r05_cfg4_stream_trace.txt.
// Serves (round 6): the stacks whose GEOMETRY is compiled (aidax_layout.h: StGeoA .. — layer count, two or three taps, power-of-two dilations
// incl. repeated cycles; conv_st_shape), blocks of 64 / 128 / 256 frames (a host's period), the fused form. The ring geometry, which taps come
// from HBM, the staging slots and the layers of each wave follow from the geometry at compile time (StG<G>).
constexpr int kStThreads = 256;
constexpr int kStRingReach = 64;                              // a tap up to this many frames back lives in the LDS ring in front of its layer; a further one comes from HBM
constexpr int kStTapVecs = 3 * 2 * 16;                        // one tap of one tile in the staging area: [term][channel half][frame]
constexpr int kStX0 = 4;                                      // frames of layer 0's input history in front of the audio row
// The chain moves in blocks of EIGHT frames, two steps per tick (a six-stage cascade then trails the tiles by two and a half ticks, not five:
// the launch ends with the post pass's last stage, alone). d1 = ticks between a tile entering the pre pass and layer 0 reading it.
constexpr int kStChainBlock = 8;
__device__ __forceinline__ int st_d1(int Kp) { return Kp / 2 + 1; }
template <int NF> __device__ __forceinline__ int st_ticks(int Kp, int Kq) { return (NF / kStChainBlock - 1 + (Kq - 1) + 2 * (st_d1(Kp) + 3)) / 2 + 1; }
constexpr int kStHandLanes = 11;                           // pre pass: lanes 0 .. 5, post pass: lanes 6 .. 11 (a pass's last stage writes the row: lane 11 has no slot)
constexpr int kStHandFloats = 2 * kStHandLanes * kStChainBlock;

// Everything a geometry G (aidax_layout.h) implies, at compile time.
template <class G> struct StG {
    static constexpr int NL = G::NL, K = G::K, KS = (G::K + 1) / 2;
    static_assert(K == 2 || K == 3, "two or three taps");
    static constexpr int shift(int l, int tap) { return (K - 1 - tap) * G::dil[l]; }       // frames back of tap `tap` (0 = the oldest)
    static constexpr int H(int l) { return (K - 1) * G::dil[l]; }                           // the layer's history: a power of two
    static constexpr bool far(int l, int tap) { return shift(l, tap) > kStRingReach; }
    static constexpr bool has_far(int l) { return far(l, 0); }
    static constexpr int ring_hist(int l)                                                   // frames of history the ring in front of layer l holds
    {
        int r = 0;
        for (int tap = 0; tap < K; ++tap) if (!far(l, tap) && shift(l, tap) > r) r = shift(l, tap);
        return r;
    }
    static constexpr int wave_of(int l) { return l < G::wbeg[1] ? 1 : l < G::wbeg[2] ? 2 : 3; }
    // tiles of the ring in front of layer l: its history, the tile in flight, one more where producer and consumer are different waves
    static constexpr int nt(int l) { return l == 0 ? 0 : (ring_hist(l) + 15) / 16 + 1 + (wave_of(l - 1) != wave_of(l) ? 1 : 0); }
    static constexpr int len(int l) { return 16 * nt(l); }
    static constexpr int off(int l) { int o = 0; for (int i = 1; i < l; ++i) o += 6 * len(i); return o; }      // in 16-byte vectors
    static constexpr int ring_vecs = off(NL);
    static constexpr int far_slot(int l, int tap)                                           // the staging slot of a far tap (layers ascending, the oldest tap first)
    {
        int n = 0;
        for (int i = 1; i < NL; ++i) for (int t = 0; t < K; ++t) { if (i == l && t == tap) return n; if (far(i, t)) ++n; }
        return n;
    }
    static constexpr int n_far = far_slot(NL, 0);
    static constexpr int stage_vecs = n_far * kStTapVecs;
    static constexpr int fetch_vecs(int l) { return (6 * ring_hist(l) + kWave - 1) / kWave; }   // per lane, a ring's history at the launch's start
    // the rings a wave asks for before the control word (at most five vectors per lane in flight) / behind the first barrier
    static constexpr int cum_fetch(int l) { int c = 0; for (int i = G::wbeg[wave_of(l) - 1]; i <= l; ++i) c += fetch_vecs(i); return c; }
    static constexpr bool early(int l) { return fetch_vecs(l) > 0 && cum_fetch(l) <= 5; }
    static constexpr bool late(int l) { return fetch_vecs(l) > 0 && cum_fetch(l) > 5; }
    static constexpr int early_slot(int l) { return cum_fetch(l) - fetch_vecs(l); }
    static constexpr bool valid()
    {
        if (G::wbeg[0] != 1 || G::wbeg[3] != NL || NL > kMaxConvLayers) return false;
        for (int l = 0; l < NL; ++l) if (G::dil[l] < 1 || (G::dil[l] & (G::dil[l] - 1))) return false;
        return (K - 1) * G::dil[0] <= kStX0;
    }
    static_assert(valid(), "geometry");
};
template <class G, int NF> __host__ __device__ constexpr size_t convst_lds_floats()
{
    return (size_t)StG<G>::ring_vecs * 4 + kStX0 + NF /* the audio row, layer 0's input history in front */
         + 20 /* Dense */ + 64 /* layer 0: [3 taps][16] + bias */ + (StG<G>::NL - 1) * 16 /* biases of the other layers */ + kStHandFloats
         + StG<G>::stage_vecs * 4 /* the far taps of the tile to come, straight from HBM (global_load_lds) */;
}

// A lane's place in a ring WALKS: frame f sits at f mod LEN, a tile later the lane reads / writes sixteen frames on — one add and a wrap
// per tile and place (computed from the tile number every time — two taps, the writer — the ring arithmetic was ~75 scalar instructions
// per tile and layer: a third of what a wave issued).
template <int LEN> __device__ __forceinline__ int st_place0(int nl, int shift) { return (nl - shift + 16 * LEN) % LEN; }
template <int LEN> __device__ __forceinline__ void st_walk(int& p) { p += 16; p = p >= LEN ? p - LEN : p; }
// Issue priority by progress (k_conv_ms's, for the same reason): a CU's four workgroups compete for its SIMDs, the arbiter favours the oldest
// wave, and the workgroup dispatched last trails the first by microseconds — the launch ends with it, the CU three quarters idle. A wave's
// priority steps down with every tick, cyclically over the four levels: of a CU's workgroups the ones behind go first.
__device__ __forceinline__ void st_prio(int tick)
{
    switch (tick & 3) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
}
// the places a lane reads layer L's k-steps from (tile 0). Three taps: k-step 0 = the oldest tap in both halves (the packed form:
// ConvLayer::ms_packed0), k-step 1 = [middle tap | newest tap]; two taps: one k-step, [oldest | newest] (p0 unused). A tap that comes from HBM
// has no place in the ring — the lane's own frame stands in.
struct StRead { int p0, p1; };
template <class G, int L> __device__ __forceinline__ StRead st_read0(int q, int nl)
{
    using S = StG<G>;
    constexpr int LEN = S::len(L);
    constexpr int s0 = (S::K == 3 && !S::far(L, 0)) ? S::shift(L, 0) : 0;
    constexpr int tl = S::K == 3 ? 1 : 0;                     // the tap in the low half of the plain k-step
    constexpr int s1 = !S::far(L, tl) ? S::shift(L, tl) : 0;
    const bool lo = q < 2;
    return StRead{ st_place0<LEN>(nl, s0), st_place0<LEN>(nl, lo ? s1 : 0) };
}
// one vector of layer L's history (a ring in HBM: the frame at time tau at index tau mod H, `pos` = the time of this block's first frame) ->
// its place in the LDS ring (v-th of the 6 hl the ring takes: strip v / hl, frame v % hl - hl)
template <class G, int L> __device__ __forceinline__ cs_u32x4 st_ring_fetch(const cs_u32x4* h, uint32_t pos, int v)
{
    constexpr int hl = StG<G>::ring_hist(L), H = StG<G>::H(L);
    if constexpr (hl == 0) return cs_u32x4{ 0u, 0u, 0u, 0u };
    else {
        if (v >= 6 * hl) return cs_u32x4{ 0u, 0u, 0u, 0u };
        const int strip = v / hl, kf = v % hl;
        return h[strip * H + (int)((pos + (uint32_t)(H - hl + kf)) & (uint32_t)(H - 1))];
    }
}
template <class G, int L> __device__ __forceinline__ void st_ring_put(cs_u32x4* pl, int v, const cs_u32x4& hv)
{
    constexpr int hl = StG<G>::ring_hist(L), LEN = StG<G>::len(L);
    if constexpr (hl != 0) {
        if (v >= 6 * hl) return;
        const int strip = v / hl, f = v % hl - hl;
        pl[StG<G>::off(L) + strip * LEN + (f + 16 * LEN) % LEN] = hv;
    }
}
// the activated outputs of layer LR - 1 (a lane: channels 4q .. 4q+3 of frame 16 t + nl) -> the ring in front of layer LR, at the lane's place po
template <class G, int LR> __device__ __forceinline__ void st_emit(cs_u32x4* pl, int q, int& po, const f32x4& v)
{
    constexpr int LEN = StG<G>::len(LR);
    cs_u32x2 tm[3];
    cs_split4(v, tm);
    cs_u32x2* dst = reinterpret_cast<cs_u32x2*>(pl + StG<G>::off(LR) + (q >> 1) * LEN + po) + (q & 1);
#pragma unroll
    for (int term = 0; term < 3; ++term) dst[term * 4 * LEN] = tm[term];
    st_walk<LEN>(po);
}
// One tile of layer L on the matrix cores: tile t of its input ring (+ the taps that come from HBM, out of the staging area) -> the activated
// outputs. Three taps: k-step 0 = [oldest tap | the same, packed], k-step 1 = [middle tap | newest tap] (conv_ms_tap); two taps: one plain k-step
// [oldest | newest]. The new history leaves from the newest tap's registers into the layer's ring in HBM (hist).
template <class G, int L, int NF, bool kEarlyReads>
__device__ __forceinline__ f32x4 st_tile(cs_u32x4* pl, const float* biasl, int t, const cs_u32x4 (&afr)[StG<G>::KS][3], int q, int nl, int activation, StRead& rd,
                                         cs_u32x4* hist, uint32_t pos, int stage_vec, int tune)
{
    using S = StG<G>;
    constexpr int K = S::K, H = S::H(L), LEN = S::len(L);
    const cs_u32x4* ring = pl + S::off(L) + (q & 1) * LEN;
    const bool lo = q < 2;
    const int i0 = rd.p0, i1 = rd.p1;
    if constexpr (K == 3) st_walk<LEN>(rd.p0);
    st_walk<LEN>(rd.p1);
    cs_u32x4 b0[3], b1[3];
    if constexpr (K == 3) {
        // k-step 0, packed: instruction i multiplies [w0 | w1], [w2 | w0], [w1 | w0] (the packer's fragments) by terms {0, 0, 1}[i] of the oldest tap in
        // the k-step's first half and {0, 1, 2}[i] in its second: a lane's three reads are term 0, then one and two terms on from where its half starts
        if constexpr (!S::far(L, 0)) {
            const int o1 = lo ? 0 : 2 * LEN;
            b0[0] = ring[i0]; b0[1] = ring[o1 + i0]; b0[2] = ring[2 * LEN + o1 + i0];
        } else {
            // ... out of the staging area ([slot][term][half][frame], where the wave's LDS-DMA put it a tick ago)
            const cs_u32x4* stg = pl + stage_vec + S::far_slot(L, 0) * kStTapVecs + (q & 1) * 16 + nl;
            const int o1 = lo ? 0 : 32;
            b0[0] = stg[0]; b0[1] = stg[o1]; b0[2] = stg[32 + o1];
        }
    }
    {
        // the plain k-step: [middle (three taps) or oldest (two) | newest]
        constexpr int tl = K == 3 ? 1 : 0;
        if constexpr (!S::far(L, tl)) {
#pragma unroll
            for (int term = 0; term < 3; ++term) b1[term] = ring[term * 2 * LEN + i1];
        } else {
            // the low half out of the staging area, the newest tap (second half) out of the ring: a per-lane base (and, unless the ring is one
            // tile long like a staging strip, a per-lane distance from term to term)
            const int a1 = lo ? stage_vec + S::far_slot(L, tl) * kStTapVecs + (q & 1) * 16 + nl : S::off(L) + (q & 1) * LEN + i1;
            if constexpr (2 * LEN == 32) {
#pragma unroll
                for (int term = 0; term < 3; ++term) b1[term] = pl[a1 + term * 32];
            } else {
                const int ts = lo ? 32 : 2 * LEN;
#pragma unroll
                for (int term = 0; term < 3; ++term) b1[term] = pl[a1 + term * ts];
            }
        }
    }
    f32x4 acc = *reinterpret_cast<const f32x4*>(biasl + (L - 1) * 16 + 4 * q);
    // (all the reads are on their way before the first product: left alone the compiler asks for a fragment right where it is used, and a
    // tile pays the LDS round trip four times in a row — the wave that sets the tick has two or three tiles per tick. Not on a wave that
    // carries three layers' A fragments: the twelve registers this costs are twelve it does not have.)
    if constexpr (kEarlyReads) __builtin_amdgcn_sched_barrier(0);
    if constexpr (K == 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cs_bf16x8, afr[0][i]), __builtin_bit_cast(cs_bf16x8, b0[i]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int th = 0; th < 3; ++th)
#pragma unroll
        for (int tw = 0; tw < 3 - th; ++tw)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cs_bf16x8, afr[S::KS - 1][tw]), __builtin_bit_cast(cs_bf16x8, b1[th]), acc, 0, 0, 0);
    // the history: the block's last H frames (every frame where a far tap of this very launch reads the block back), as the vectors they are
    // (lanes q >= 2 of the plain k-step: term, half q & 1, frame 16 t + nl), frame f at index (pos + f) mod H of the layer's ring in HBM
    int f = 16 * t + nl;
    asm volatile("" : "+v"(f));                               // (opaque: otherwise every layer's store address becomes a 64-bit pointer that is carried, and stepped, through every tick)
    constexpr bool all = S::has_far(L) || H >= NF;
    if ((all || 16 * t + 15 >= NF - H) && !lo && (all || f >= NF - H) && !(tune & 32768)) {      // (bit 32768, test build: no history leaves — what the stores cost; wrong state)
        const int j = (int)((pos + (uint32_t)f) & (uint32_t)(H - 1));
#pragma unroll
        for (int term = 0; term < 3; ++term) hist[(term * 2 + (q & 1)) * H + j] = b1[term];
    }
    return cs_activate(acc, activation);
}

template <int I, int N, class F> __device__ __forceinline__ void st_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); st_for<I + 1, N>(f); }
}

// G: the stack's geometry; NF: the block, 64 / 128 / 256 frames (a host's period: the reference's run() gets the host's block, rt-neural-generic.cpp:484);
// kTanh: every layer's activation is tanh (what a WaveNet-like stack has): the epilogue's switch on the layer's activation — a dozen scalar
// instructions and branches per tile and layer — is compiled out.
template <class G, int NF, bool kTanh>
__global__ __launch_bounds__(kStThreads, 4) void k_conv_st(LaunchArgs a, ConvDesc d)
{
    using S = StG<G>;
    constexpr int NL = S::NL, K = S::K, KS = S::KS, NT = NF / 16;
    static_assert(NF % 64 == 0 && NF <= kConvsFrames, "whole tiles, whole float4 rows");
    static_assert(convst_lds_floats<G, NF>() * 4 <= 40 * 1024, "four workgroups per CU");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sg = blockIdx.x;
    // (Which role on which wave does not matter for the SIMDs' balance: the CU rotates the wave -> SIMD assignment from one of its
    // workgroups to the next — every SIMD carries one wave of each role as it is. Rotating the roles as well cancels that: 62.9 against
    // 51.8 us, scratch/st_trace.py.)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, nl = lane & 15;
    constexpr int n = NF;
    const size_t rstride = a.row_stride ? a.row_stride : (size_t)n;
    cs_u32x4* pl = reinterpret_cast<cs_u32x4*>(smem);
    float* xbuf = smem + (size_t)S::ring_vecs * 4;
    float* xrow = xbuf + kStX0;
    float* wdl = xrow + NF;                                   // Dense weights [16] + bias
    float* l0w = wdl + 20;                                    // layer 0: [tap][16] (tanh: times 2 log2 e), then its bias at [48]
    float* biasl = l0w + 64;                                  // [layer - 1][16]
    float* hand = biasl + (NL - 1) * 16;
    const float* W = a.wpack;
    float* st_base = a.nn + (size_t)sg * a.nn_stride;
    const StreamCtl& ctl = a.ctl[sg];
    StreamState& st = a.st[sg];
    const float* in_row = a.in + (size_t)sg * rstride;
    float* out_row = a.out + (size_t)sg * rstride;

#ifdef AIDAX_CONV_TRACE
    // measurement build (scratch/st_trace.py): per wave and tick, (cycles from the kernel's start to the tick's work) / 16 << 16 | cycles of work,
    // in 256 words behind the histories; the chain wave copies them into the output row at the end
    uint32_t* trace = reinterpret_cast<uint32_t*>(st_base + d.st_trace_off);
    const unsigned long long t_start = clock64();
    const unsigned long long w_start = wall_clock64();
    unsigned long long t_tick = t_start;
    { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); if (lane == 0) trace[128 + wave] = hwid; }
#define ST_TICK_BEGIN() do { t_tick = clock64(); } while (0)
#define ST_TICK_END(tick) do { const unsigned long long now = clock64(); if (lane == 0) trace[wave * 32 + (tick)] = (uint32_t)(((t_tick - t_start) >> 4) << 16) | (uint32_t)((now - t_tick) & 0xffff); if ((tick) + 1 == T) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#else
#define ST_TICK_BEGIN() do { } while (0)
#define ST_TICK_END(tick) do { } while (0)
#endif
    // Everything the launch reads from memory before its first tick is requested HERE, in front of the control word the first branch depends
    // on — reads only, whatever that branch decides. Wave 0: the audio row, the smoothers, both passes' coefficients and state by lane (a lane
    // beyond its cascade holds some stage's numbers and does not run); waves 1 .. 3: the time the histories' rings stand at, then the histories
    // of their first rings, their layers' fragments (what a wave needs only at its first tile — a tick or more away — is asked for behind the
    // first barrier: see the roles).
    auto hist_of = [&](int l) { return reinterpret_cast<cs_u32x4*>(st_base + d.L[l].ms_state_off); };
    auto fetch_afrags = [&](int l, cs_u32x4 (&afr)[KS][3]) {
        const cs_u32x4* rec = reinterpret_cast<const cs_u32x4*>(W + d.L[l].ms_w_off) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 3; ++t) afr[ks][t] = rec[(ks * 3 + t) * kWave];
    };
    const bool isQ = lane >= 6;
    const int stage = isQ ? lane - 6 : lane;
    const int slot = isQ ? post_slot(stage < 6 ? stage : 0) : pre_slot(stage < 6 ? stage : 0);
    if (wave == 0) {
        float4 rowv = float4{ 0.f, 0.f, 0.f, 0.f };
        ChainPass c;
        uint32_t pending = 0;
        float pre_mem = 0.f, master_mem = 0.f, pre_tgt = 0.f, master_tgt = 0.f, pre_target = 0.f, master_target = 0.f, ramp_coef = 0.f;
        if (NF == 256 || lane < NF / 4) rowv = reinterpret_cast<const float4*>(in_row)[lane];
        chain_load(c, ctl, st, slot, false);
        pending = st.pending;
        pre_mem = st.pre_mem; master_mem = st.master_mem; pre_tgt = st.pre_tgt; master_tgt = st.master_tgt;
        pre_target = ctl.pre_target; master_target = ctl.master_target;
        ramp_coef = isQ ? ctl.master_coef : ctl.pre_coef;
        // (param_targets' reads: its verdict is written at the end of the launch — nothing in here waits for it)
        const float pt0 = ctl.p_target[0], pt1 = ctl.p_target[1], p_den = ctl.p_den;
        float ptg0 = st.p_tgt[0], ptg1 = st.p_tgt[1];
        const float pm0 = st.p_mem[0], pm1 = st.p_mem[1];
        const uint32_t flags = ctl.flags;
        if (!(flags & CTL_ENABLED) || !(flags & CTL_NET_ON)) {
            // pre-run / bypass / the model out of circuit: the chain alone, k_conv_ms's own path for such a stream
#ifdef AIDAX_CONV_TRACE
            return;                                               // (measurement build: every stream is in circuit, and the chain helpers' stamp area would cost the fourth workgroup per CU)
#endif
            // (the blocked passes' hand-over slots — kChainHandFloats, more than this kernel's own — in the ring area: no ring is in use here)
            static_assert((size_t)S::ring_vecs * 4 >= (size_t)kChainHandFloats, "the chain-only path borrows the rings");
            float* wide_hand = smem;
            ChainCtx ctx = chain_prologue<true>(ctl, st, in_row, out_row, xrow, n, lane, wide_hand);
            if (!ctx.live) return;
            chain_epilogue(ctl, st, ctx, out_row, xrow, n, lane, wide_hand);
            return;
        }
        const int Kp = (flags & CTL_EQ_PRE) ? 6 : 1, Kq = (flags & CTL_EQ_POST) ? 6 : 1;
        const int d1 = st_d1(Kp), T = st_ticks<NF>(Kp, Kq);
        // ---- the chain wave: lanes 0 .. 5 the pre pass's stages, lanes 6 .. 11 the post pass's (chain_prologue / chain_epilogue, side by side)
        if (pending & PEND_ACTIVATE) { pre_mem = pre_tgt; master_mem = master_tgt; pending &= ~PEND_ACTIVATE; }
        pre_tgt = pre_target;
        master_tgt = master_target;
        c.K = isQ ? Kq : Kp;
        c.gain_lane = isQ ? Kq - 1 : 0;
        c.active = stage == 0 ? (flags & (isQ ? CTL_DC_ON : CTL_LPF_ON)) != 0 : ((flags & CTL_EQ_BANDPASS) ? slot == BQ_MID : true);
        c.g.arm(isQ ? master_mem : pre_mem, isQ ? master_tgt : pre_tgt, ramp_coef);
        const bool run = lane < 12 && stage < c.K;
        if (NF == 256 || lane < NF / 4) reinterpret_cast<float4*>(xrow)[lane] = rowv;
        const double z1o = c.z1, z2o = c.z2;
        ExpRamp g = c.g;
        const bool is_gain = stage == c.gain_lane;
        if (!is_gain) { g.mem = 1.f; g.coef = 1.f; g.tc = 0.f; }
        const bool last = stage == c.K - 1;
        const bool fussy = run && (!c.active || g.mem * g.coef + g.tc != g.mem);
        const bool plain = __builtin_amdgcn_ballot_w64(fussy) == 0;
        const int m0 = isQ ? 2 * (d1 + 3) : 0;
        cs_lds_barrier();                                     // (the other waves' staging)
        for (int tick = 0; tick < T; ++tick) {
            if (!(AIDAX_TUNE(a) & 2048)) st_prio(tick);         // (this wave — the longest of the four — always on top instead: 42.8 against 42.3 us)
            ST_TICK_BEGIN();
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                if (plain) chain_macro_step<true, kStChainBlock, kStHandLanes>(c, g, stage, run, last, xrow, hand, NF / kStChainBlock, 2 * tick + hs - m0, lane);
                else chain_macro_step<false, kStChainBlock, kStHandLanes>(c, g, stage, run, last, xrow, hand, NF / kStChainBlock, 2 * tick + hs - m0, lane);
            }
            ST_TICK_END(tick);
            cs_lds_barrier();
        }
        if (is_gain) c.g = g;
        if (!run || !c.active) { c.z1 = z1o; c.z2 = z2o; }
        if (run) { st.z[slot][0] = c.z1; st.z[slot][1] = c.z2; }
        pre_mem = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.g.mem), 0));
        master_mem = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.g.mem), 6 + Kq - 1));
        if (NF == 256 || lane < NF / 4) reinterpret_cast<float4*>(out_row)[lane] = reinterpret_cast<const float4*>(xrow)[lane];
#ifdef AIDAX_CONV_TRACE
        if constexpr (NF == 256) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int i = lane; i < 132; i += kWave) out_row[i] = __builtin_bit_cast(float, trace[i]);
        if (lane == 0) { out_row[160] = __builtin_bit_cast(float, (uint32_t)(clock64() - t_start)); out_row[166] = __builtin_bit_cast(float, (uint32_t)blockIdx.x); out_row[161] = __builtin_bit_cast(float, (uint32_t)T);
                         out_row[162] = __builtin_bit_cast(float, (uint32_t)(w_start & 0xffffffffu)); out_row[163] = __builtin_bit_cast(float, (uint32_t)(wall_clock64() & 0xffffffffu));
                         unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                         out_row[164] = __builtin_bit_cast(float, hw); out_row[165] = __builtin_bit_cast(float, xcc); }
        }
#endif
        if (lane == 0) {
            // run() :634-640 for a model without PARAM inputs (param_targets, aidax_device.h): the smoothers' targets follow the controls,
            // the first run after a load snaps them
            if (__builtin_fabsf(ptg0 - pt0) >= FLT_EPSILON) { st.p_tgt[0] = pt0; st.p_step[0] = (pt0 - pm0) / p_den; ptg0 = pt0; }
            if (__builtin_fabsf(ptg1 - pt1) >= FLT_EPSILON) { st.p_tgt[1] = pt1; st.p_step[1] = (pt1 - pm1) / p_den; ptg1 = pt1; }
            if (pending & PEND_PARAM_FIRST) { pending &= ~PEND_PARAM_FIRST; st.p_mem[0] = ptg0; st.p_mem[1] = ptg1; }
            st.pre_mem = pre_mem; st.master_mem = master_mem;
            st.pre_tgt = pre_tgt; st.master_tgt = master_tgt;
            st.pending = pending;
        }
        return;
    }

    // ---- waves 1 .. 3: the layers. Each is self-contained from here (no value of one role alive in another's code): its early reads, the
    // control word, its share of the small staging — layer 0's input history and weights, the biases, the Dense: L2-resident weights —, its
    // ticks. A layer wave loads the histories of ITS rings — layer l's last ring_hist(l) frames out of the layer's ring in HBM — and puts them
    // into the LDS rings in front of its first tile, one to three ticks into the launch: the first rings' (up to five vectors per lane) are
    // asked for with everything else, the rest behind the first barrier (all at once, the 18 MB cfg4's launch reads at its start kept the first
    // tick waiting for 4.5 us).
    uint32_t* pos_slot = reinterpret_cast<uint32_t*>(st_base + d.ms_pos_off);
    const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)*pos_slot);
    auto out_of_circuit = [&](uint32_t flags) { return !(flags & CTL_ENABLED) || !(flags & CTL_NET_ON); };      // (wave 0 runs such a stream alone)
    const int u = (wave - 1) * kWave + lane;                  // 0 .. 191
    float sm_x = 0.f, sm_wd = 0.f, sm_l0 = 0.f, sm_b = 0.f;
    {                                                         // (requested with the rest of the early reads; stored behind the control word)
        const ConvLayer& L0 = d.L[0];
        if (u < L0.hist) sm_x = st_base[L0.ms_state_off + u];
        if (u < 17) sm_wd = u < 16 ? W[d.wd_off + u] : W[d.bd_off];
        if (u >= 64 && u < 128) {
            const int j = u - 64;
            const float scale = L0.activation == 1 ? kTwoLog2e : 1.0f;
            sm_l0 = j < 16 * K ? scale * W[L0.w_off + j] : j >= 48 ? W[L0.bs_off + j - 48] : 0.f;
        }
        if (u < (NL - 1) * 16) sm_b = W[d.L[1 + (u >> 4)].bs_off + (u & 15)];
    }
    auto stage_small = [&]() {
        if (u < d.L[0].hist) xbuf[kStX0 - d.L[0].hist + u] = sm_x;
        if (u < 17) wdl[u] = sm_wd;
        if (u >= 64 && u < 128) l0w[u - 64] = sm_l0;
        if (u < (NL - 1) * 16) biasl[u] = sm_b;
    };
    constexpr int stage_vec = (int)((convst_lds_floats<G, NF>() - (size_t)S::stage_vecs * 4) / 4);
    static_assert((convst_lds_floats<G, NF>() - (size_t)S::stage_vecs * 4) % 4 == 0, "the staging area starts on a vector");

    auto role = [&](auto Wc) {
        constexpr int Wv = decltype(Wc)::value;               // 1 .. 3
        constexpr int LB = G::wbeg[Wv - 1], LE = G::wbeg[Wv], NW = LE - LB;
        static_assert(NW >= 1 && NW <= 4, "layers per wave");
        constexpr bool kEarlyReads = NW * KS <= 4;            // (a wave with three layers of two k-steps has no registers for it)
        constexpr bool first = Wv == 1, lastw = Wv == 3;
        cs_u32x4 hv[6], afr[NW][KS][3];
        // early reads
        st_for<0, NW>([&](auto ic) {
            constexpr int i = decltype(ic)::value, L = LB + i;
            if constexpr (S::early(L)) {
#pragma unroll
                for (int r = 0; r < S::fetch_vecs(L); ++r) hv[S::early_slot(L) + r] = st_ring_fetch<G, L>(hist_of(L), pos, lane + r * kWave);
            }
        });
        st_for<0, NW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i == 0 || NW <= 2) fetch_afrags(LB + i, afr[i]);
        });
        const uint32_t flags = ctl.flags;
        if (out_of_circuit(flags)) return;
        const int Kp = (flags & CTL_EQ_PRE) ? 6 : 1, Kq = (flags & CTL_EQ_POST) ? 6 : 1, d1 = st_d1(Kp), T = st_ticks<NF>(Kp, Kq);
        stage_small();
        const ConvLayer& L0 = d.L[0];
        int act[NW];
        StRead rd[NW];
        int po[NW];                                           // this lane's place in the ring BEHIND layer LB + i (the last wave's last: unused, the Dense takes it)
        st_for<0, NW>([&](auto ic) {
            constexpr int i = decltype(ic)::value, L = LB + i;
            act[i] = kTanh ? 1 : d.L[L].activation;
            rd[i] = st_read0<G, L>(q, nl);
            po[i] = nl;
        });
        const int act0 = kTanh ? 1 : L0.activation;
        int po1 = nl;                                         // (wave 1: layer 0's place in the ring in front of layer 1)
        // the far taps of this wave's layers, from HBM into the staging area without passing through registers (LDS-DMA: the low 32 lanes'
        // (half, frame) vectors land at base + 16 lane, a term per instruction), requested when the tile before has been read: frame f - s of
        // a layer's ring in HBM — written by the block before where it is older than this block, by this very launch (st_tile) where it is not
        auto fetch_g = [&](int t) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the reads of the tile before have left the staging area)
            if (q < 2 && t < NT) {
                st_for<0, NW>([&](auto ic) {
                    constexpr int L = LB + decltype(ic)::value, H = S::H(L);
                    st_for<0, K>([&](auto tc) {
                        constexpr int tap = decltype(tc)::value;
                        if constexpr (S::far(L, tap)) {
                            int j = (int)((pos + (uint32_t)(16 * t + nl) - (uint32_t)S::shift(L, tap)) & (uint32_t)(H - 1));
                            if (AIDAX_TUNE(a) & 65536) j = nl;      // (bit 65536, test build: every far tap reads the same sixteen frames — what their trip to HBM costs; wrong output)
                            const cs_u32x4* src = hist_of(L) + (q & 1) * H + j;
#pragma unroll
                            for (int term = 0; term < 3; ++term) {
                                unsigned keep;
                                const unsigned dst = (unsigned)(size_t)(pl + stage_vec + S::far_slot(L, tap) * kStTapVecs + term * 32);      // (LDS byte address: the low 32 bits of the pointer)
                                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                             : "=&s"(keep) : "v"(src + term * 2 * H), "s"(dst) : "memory");
                            }
                        }
                    });
                });
            }
        };
        constexpr bool wave_far = [] { bool f = false; for (int l = LB; l < LE; ++l) f = f || S::has_far(l); return f; }();
        const float* wdq = wdl + 4 * q;
        cs_lds_barrier();
        // the first rings' histories go in as they arrive
        st_for<0, NW>([&](auto ic) {
            constexpr int L = LB + decltype(ic)::value;
            if constexpr (S::early(L)) {
#pragma unroll
                for (int r = 0; r < S::fetch_vecs(L); ++r) st_ring_put<G, L>(pl, lane + r * kWave, hv[S::early_slot(L) + r]);
            }
        });
        if constexpr (NW >= 3) fetch_afrags(LB + 1, afr[1]);
        // ... the first of the others is asked for now, behind the first barrier, and goes in right in front of the wave's first tile
        constexpr int first_late = [] { for (int l = LB; l < LE; ++l) if (S::late(l)) return l; return -1; }();
        if constexpr (first_late >= 0) {
#pragma unroll
            for (int r = 0; r < S::fetch_vecs(first_late); ++r) hv[r] = st_ring_fetch<G, first_late>(hist_of(first_late), pos, lane + r * kWave);
        }
        if constexpr (wave_far) fetch_g(0);
        int tick = 0;
        for (; tick < d1 + (Wv - 1); ++tick) cs_lds_barrier();     // (nothing to do yet: the pipeline fills)
        if constexpr (first_late >= 0) {
#pragma unroll
            for (int r = 0; r < S::fetch_vecs(first_late); ++r) st_ring_put<G, first_late>(pl, lane + r * kWave, hv[r]);
        }
        st_for<0, NW>([&](auto ic) {                          // (a geometry with more late rings per wave than one: one after the other, the wave waits)
            constexpr int L = LB + decltype(ic)::value;
            if constexpr (S::late(L) && L != first_late) {
#pragma unroll
                for (int r = 0; r < S::fetch_vecs(L); ++r) hv[r] = st_ring_fetch<G, L>(hist_of(L), pos, lane + r * kWave);
#pragma unroll
                for (int r = 0; r < S::fetch_vecs(L); ++r) st_ring_put<G, L>(pl, lane + r * kWave, hv[r]);
            }
        });
        st_for<2, NW>([&](auto ic) {                          // (in the registers the history just left; L2-resident, two tiles of work in front of its first use)
            constexpr int i = decltype(ic)::value;
            if constexpr (NW >= 3) fetch_afrags(LB + i, afr[i]);
        });
#pragma nounroll
        for (; tick < T; ++tick) {
            const int t = tick - d1 - (Wv - 1);
            if (!(AIDAX_TUNE(a) & 2048)) st_prio(tick);
            ST_TICK_BEGIN();
            if (t < NT) {
                if constexpr (wave_far) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the staged taps have landed
                if constexpr (first) {
                    // ---- layer 0 (one scalar input, fp32 FMAs). The tile of the model's input: x * in_gain, in place (the Dense's skip path and
                    // layer 0's history read it scaled)
                    if (lane < 16) xrow[16 * t + lane] = xrow[16 * t + lane] * a.in_gain;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    f32x4 v = *reinterpret_cast<const f32x4*>(l0w + 48 + 4 * q);
#pragma unroll
                    for (int tap = 0; tap < K; ++tap) {
                        const float x = xrow[16 * t + nl - (K - 1 - tap) * G::dil[0]];
                        const f32x4 w = *reinterpret_cast<const f32x4*>(l0w + tap * 16 + 4 * q);
                        v.x = __builtin_fmaf(w.x, x, v.x); v.y = __builtin_fmaf(w.y, x, v.y);
                        v.z = __builtin_fmaf(w.z, x, v.z); v.w = __builtin_fmaf(w.w, x, v.w);
                    }
                    constexpr int h0 = (K - 1) * G::dil[0];
                    if (t == NT - 1 && lane >= 16 - h0 && lane < 16) st_base[L0.ms_state_off + lane - (16 - h0)] = xrow[NF - 16 + lane];
                    st_emit<G, 1>(pl, q, po1, cs_activate(v, act0));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                st_for<0, NW>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, L = LB + i;
                    const f32x4 v = st_tile<G, L, NF, kEarlyReads>(pl, biasl, t, afr[i], q, nl, act[i], rd[i], hist_of(L), pos, stage_vec, AIDAX_TUNE(a));
                    if constexpr (L + 1 < NL) {
                        st_emit<G, L + 1>(pl, q, po[i], v);
                        if constexpr (i + 1 < NW) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    } else {
                        // ---- Dense(16, 1) + skip / output gain (:171-181; emit()'s arithmetic in k_conv_ms)
                        if constexpr (wave_far) fetch_g(t + 1);
                        float y = wdq[0] * v.x;
                        y = __builtin_fmaf(wdq[1], v.y, y);
                        y = __builtin_fmaf(wdq[2], v.z, y);
                        y = __builtin_fmaf(wdq[3], v.w, y);
                        const Pair r2 = share_rows(y);
                        y = r2.lo + r2.hi;
                        const Pair r4 = share_halves(y);
                        y = (r4.lo + r4.hi) + wdl[16];
                        const int f = 16 * t + nl;
                        const float x = xrow[f];
                        const float o = (a.input_skip ? x + y : y) * a.out_gain;
                        __builtin_amdgcn_wave_barrier();
                        if (q == 0) xrow[f] = o;
                    }
                });
                if constexpr (wave_far && !lastw) fetch_g(t + 1);
            }
            ST_TICK_END(tick);
            cs_lds_barrier();
        }
        if constexpr (lastw) {
            if (lane == 0) *pos_slot = (pos + (uint32_t)NF) % d.ms_pos_mod;      // the stream's time moves on by the block
        }
    };
    if (wave == 1) role(std::integral_constant<int, 1>{});
    else if (wave == 2) role(std::integral_constant<int, 2>{});
    else role(std::integral_constant<int, 3>{});
}

size_t convs_lds_bytes() { return convs_lds_floats() * sizeof(float); }

// ---- k_conv_st's instantiations: geometry x block length x (every layer tanh?)
typedef void (*StKernel)(LaunchArgs, ConvDesc);
struct StEntry { StKernel fn; size_t lds; };
template <class G, int NF> static StEntry st_entry_nf(bool all_tanh)
{
    return StEntry{ all_tanh ? k_conv_st<G, NF, true> : k_conv_st<G, NF, false>, convst_lds_floats<G, NF>() * sizeof(float) };
}
// the streaming kernel of geometry `geo` for blocks of n_frames, {nullptr, 0} where there is none (a block that is not 64, 128 or 256 frames)
static StEntry st_entry(int geo, uint32_t n_frames, bool all_tanh)
{
    return st_geo_dispatch(geo, [&](auto g) {
        using G = decltype(g);
        switch (n_frames) {
        case 64: return st_entry_nf<G, 64>(all_tanh);
        case 128: return st_entry_nf<G, 128>(all_tanh);
        case 256: return st_entry_nf<G, 256>(all_tanh);
        default: return StEntry{ nullptr, 0 };
        }
    });
}
bool conv_st_block_ok(uint32_t n_frames) { return n_frames == 64 || n_frames == 128 || n_frames == 256; }

int convs_resident_streams(int device, int st_geo)
{
    int per_cu = 0, cus = 0;
    const void* fn = reinterpret_cast<const void*>(k_conv_ms<true, false>);      // (both instantiations have the same footprint)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kCsThreads, convs_lds_bytes()) != hipSuccess) return 0;
    if (st_geo >= 0) {                                        // whole-tile blocks go through k_conv_st: both must be resident at the pool's size
        const StEntry e = st_entry(st_geo, 256, false);
        int st = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&st, reinterpret_cast<const void*>(e.fn), kStThreads, e.lds) != hipSuccess) return 0;
        if (st < per_cu) per_cu = st;
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu * cus;
}

// fused: the launch is the whole run() of every stream (a.mode must be MODE_CHAIN); otherwise applyModel only. n_frames <= 256.
hipError_t launch_conv_ms_kernel(const LaunchArgs& a, const ConvDesc& d, bool fused, hipStream_t stream)
{
    if (!d.ms_ok || a.n_frames > (uint32_t)kConvsFrames || (fused && a.mode != MODE_CHAIN)) return hipErrorInvalidValue;
    const bool full = a.n_frames == (uint32_t)kConvsFrames;
    // the streaming form: a stack with a compiled geometry, a block of 64 / 128 / 256 frames, rows the kernel's 16-byte accesses can take
    // (a time slice of a longer block keeps the block's pitch: any multiple of four frames)
    if (fused && d.st_ok && conv_st_block_ok(a.n_frames) && (a.row_stride & 3u) == 0) {
        bool all_tanh = true;
        for (int l = 0; l < d.n_layers; ++l) all_tanh = all_tanh && d.L[l].activation == 1;
        const StEntry e = st_entry(d.st_ok - 1, a.n_frames, all_tanh);
        hipLaunchKernelGGL(e.fn, dim3(a.n_streams), dim3(kStThreads), e.lds, stream, a, d);
        return hipGetLastError();
    }
    typedef void (*Fn)(LaunchArgs, ConvDesc);
    const Fn fn = fused ? (full ? k_conv_ms<true, true> : k_conv_ms<true, false>) : (full ? k_conv_ms<false, true> : k_conv_ms<false, false>);
    hipLaunchKernelGGL(fn, dim3(a.n_streams), dim3(kCsThreads), convs_lds_bytes(), stream, a, d);
    return hipGetLastError();
}

}  // namespace aidax
