// aidax_mfma.hip — k_mfma: recurrent layers too wide for one wave's registers (stacked LSTM/GRU,
// hidden 96/128 ...: BASELINE config #5) on the matrix cores.
//
// The per-sample work of such a model IS a contraction once streams are batched: for every
// frame, gates[4H] = [W|U] . [x ; h(t-1)] + b for each stream. One workgroup (4 or 8 waves) takes
// kMfmaStreams = 16 streams as the N dimension of v_mfma_f32_16x16x4_f32 (exact fp32 in, fp32 acc):
//
//   A (16 rows x 4 k)   weights, pre-packed on the host as ready fragments (aidax_pack.cpp:pack_mfma):
//                       a tile's 16 rows are 4 units x their 4 gate rows, so the accumulator a lane
//                       ends up with (rows 4a..4a+3, column n) is exactly i,f,g,o (or z,r,n_h,n_x) of
//                       unit 4T+a for stream n: the cell update needs no cross-lane traffic at all.
//   B (4 k x 16 streams) the input vector, kept TRANSPOSED in LDS ([k][stream]) so a fragment is the
//                       64 consecutive floats starting at 64*kk: one conflict-free ds_read_b32.
//   bias                a [unit][4] table in LDS the accumulators start from.
//
// Layers are skewed in time: at tick T layer l works on frame T-l and the Dense on frame T-L, so
// everything a tick reads was written in the tick before (h buffers and the input column are
// double-buffered by tick parity) and ONE workgroup barrier per tick suffices.
//
// Weights stream from L2 every tick (437 KiB for LSTM-96 x2; each fragment load serves 16 streams),
// requested two k-step groups ahead of the MFMAs that use them. The DSP chain around the model runs
// in the packed k_chain launches of aidax_kernels.hip (split form), this kernel is applyModel only.
//
// Measured (MI355X, LSTM-96 x2, 256 frames): 2.50 ms per block for up to 4096 streams
// (k_stack, the VALU form: 5.1 ms at 2048), i.e. 23 k cycles per tick against 13.8 k of pure
// matrix-core time (432 MFMAs x 32 cycles per SIMD); the rest is the cell update (VALU next to
// MFMAs is not hidden on this machine, scratch/umfma.hip), L2 latency and the barrier.
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaChunk = 256;          // frames staged in LDS at a time

__host__ __device__ inline size_t mfma_lds_floats(int hidden, int n_layers, int n_frames)
{
    const size_t nP = (size_t)(((n_frames < kMfmaChunk ? n_frames : kMfmaChunk) + 3) & ~3);
    return (size_t)kMfmaStreams * nP                          /* xb: audio rows            */
         + 2 * 64                                             /* xin[parity][4][n]         */
         + (size_t)n_layers * 2 * hidden * kMfmaStreams       /* hT[l][parity][unit][n]    */
         + (size_t)n_layers * hidden * kMfmaStreams           /* cT[l][unit][n]            */
         + (size_t)n_layers * hidden * 4                      /* bias[l][unit][4 rows]     */
         + (size_t)((hidden + 1 + 3) & ~3)                    /* Dense weights + bias      */
         + kMfmaStreams;                                      /* live flags                */
}

__device__ __forceinline__ float row_sum16(float v)
{
    v = v + dpp_take<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = v + dpp_take<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = v + dpp_take<0x141, 0xf>(v);    // row_half_mirror
    v = v + dpp_take<0x140, 0xf>(v);    // row_mirror
    return v;
}

template <int TPW>
__device__ __forceinline__ void load_frag(f32x4 (&dst)[TPW], const f32x4* __restrict__ p)
{
#pragma unroll
    for (int q = 0; q < TPW; ++q) dst[q] = p[q];
}

// One recurrent layer's gate pre-activations for this wave's TPW tiles:
//   acc[tl] += A(group g) * B(group g) for the layer's `g_in + g_rec` groups of four k-steps, where
//   B = below[64*kk + lane] for the first g_in groups (h of the layer below) and own[...] after that.
// The A fragments of later groups are requested before the 4*TPW MFMAs of group g are issued: left to
// itself the compiler loads each float4 right in front of its four MFMAs and exposes the L2 latency
// 6 times per group (measured 36 k cycles per tick against 14 k of matrix-core time).
template <int TPW>
__device__ __forceinline__ void gates_mfma(f32x4 (&acc)[TPW], const f32x4* __restrict__ ap, const float* below,
                                           const float* own, int g_in, int g_tot, int lane)
{
    // Fragments are requested two groups ahead (one group is ~400-800 cycles of MFMAs, the L2 round trip
    // under load is of that order). Three buffers rotate by NAME through a 3x unrolled body: a rotating
    // copy costs TPW*4 v_mov per group, and VALU work next to MFMAs is not hidden on this machine
    // (scratch/umfma.hip: +4.7 cycles per VALU instruction issued between two MFMAs of one wave).
    f32x4 f0[TPW], f1[TPW], f2[TPW];
    const int stride = kWave * TPW;
    const int last = g_tot - 1;
    auto fetch = [&](f32x4 (&f)[TPW], int g) {
        load_frag<TPW>(f, ap + (size_t)(g < last ? g : last) * stride);      // the tail re-reads the last group
    };
    // B fragments (LDS) run one group ahead as well: read for group g+1, then the MFMAs of group g
    float bc[4], bn[4];
    auto read_b = [&](float (&b)[4], int g) {
        const int gg = g < last ? g : last;
        const float* src = gg < g_in ? below + 256 * gg : own + 256 * (gg - g_in);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = src[64 * j + lane];
    };
    auto group = [&](const f32x4 (&f)[TPW], const float (&b)[4]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                const int e = j * TPW + tl;
                acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[e / 4][e % 4], b[j], acc[tl], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    load_frag<TPW>(f0, ap);
    load_frag<TPW>(f1, ap + (size_t)(last > 0 ? 1 : 0) * stride);
    read_b(bc, 0);
    auto roll = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) bc[j] = bn[j];
    };
    int g = 0;
    for (; g + 3 <= g_tot; g += 3) {
        fetch(f2, g + 2); read_b(bn, g + 1); group(f0, bc); roll();
        fetch(f0, g + 3); read_b(bn, g + 2); group(f1, bc); roll();
        fetch(f1, g + 4); read_b(bn, g + 3); group(f2, bc); roll();
    }
    if (g < g_tot) {                                   // one or two groups left, buffers in the same rotation
        read_b(bn, g + 1); group(f0, bc);
        if (g + 1 < g_tot) group(f1, bn);
    }
}

// RES: a single recurrent layer whose A fragments a wave can keep in registers for the whole launch
// ((hidden/4) * TPW floats per lane: 8 for 32 units, 100 for 80, 128 for 128): no L2 fragment stream and no
// address arithmetic in the frame loop.
template <int TPW, int NW, bool RES>
__global__ __launch_bounds__(NW * kWave) void k_mfma(LaunchArgs a, MfmaDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 4 * TPW * NW;
    constexpr int kMfmaThreads = NW * kWave, kMfmaWaves = NW;
    constexpr int NS = kMfmaStreams;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int NL = d.n_layers;
    const int Ht = d.hidden_true;          // the model's width (H is that rounded up to 16)
    const int I = a.input_size;
    const int s_base = blockIdx.x * NS;
    const int chunk = n < kMfmaChunk ? n : kMfmaChunk;
    const int nP = (chunk + 3) & ~3;

    float* xb   = smem;                                   // [NS][nP]
    float* xin  = xb + NS * nP;                           // [2][4][NS]
    float* hT   = xin + 2 * 64;                           // [NL][2][H][NS]
    float* cT   = hT + (size_t)NL * 2 * H * NS;           // [NL][H][NS]
    float* bl   = cT + (size_t)NL * H * NS;               // [NL][H][4]
    float* wdl  = bl + (size_t)NL * H * 4;                // Dense weights, bias at [H]
    float* livef = wdl + ((H + 1 + 3) & ~3);              // [NS]

    const float* W = a.wpack;
    const int mode = a.mode;

    // ---- per-stream bookkeeping: lanes tid < NS own stream s_base+tid (live flag, PARAM smoothers)
    float p_mem[2] = { 0.f, 0.f }, p_tgt[2] = { 0.f, 0.f }, p_step[2] = { 0.f, 0.f };
    uint32_t pending = 0;
    bool mine_live = false;
    if (tid < NS) {
        const int sg = s_base + tid;
        const bool valid = sg < (int)a.n_streams;
        if (valid) {
            StreamState& st = a.st[sg];
            p_mem[0] = st.p_mem[0]; p_mem[1] = st.p_mem[1];
            p_tgt[0] = st.p_tgt[0]; p_tgt[1] = st.p_tgt[1];
            p_step[0] = st.p_step[0]; p_step[1] = st.p_step[1];
            pending = st.pending;
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
    for (int i = tid; i < H + 1; i += kMfmaThreads) wdl[i] = W[d.wd_off + i];
    for (int l = 0; l < NL; ++l)
        for (int i = tid; i < H * 4; i += kMfmaThreads) bl[l * H * 4 + i] = W[d.L[l].b_off + i];
    // recurrent state -> LDS (parity 0 is what tick 0 reads)
    for (int l = 0; l < NL; ++l) {
        const MfmaLayer& L = d.L[l];
        for (int i = tid; i < H * NS; i += kMfmaThreads) {
            const int u = i / NS, sn = i % NS, sg = s_base + sn;
            const bool valid = sg < (int)a.n_streams;
            const float* stp = a.nn + (size_t)(valid ? sg : 0) * a.nn_stride + L.state_off;
            hT[((size_t)l * 2 + 0) * H * NS + i] = (valid && u < Ht) ? stp[u] : 0.f;         // padded units rest at 0
            cT[(size_t)l * H * NS + i] = (valid && u < Ht && L.cell == 0) ? stp[Ht + u] : 0.f;
        }
    }
    __syncthreads();

    const int l_first = (NW == 8 && wave >= 4) ? NL / 2 : 0;       // waves w and w+4 share a SIMD (scratch/umfma.hip)
    // layer 0's model-input k-step (x, PARAM1, PARAM2, 0): loop-invariant, kept in registers
    float w_in0[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) w_in0[tl] = W[d.L[0].w_in_off + ((size_t)wave * kWave + lane) * TPW + tl];
    constexpr int G = H / 16;                              // k-step groups of the recurrent part
    f32x4 wres[RES ? G * TPW : 1];
    if constexpr (RES) {
        const f32x4* ap = reinterpret_cast<const f32x4*>(W + d.L[0].w_big_off) + ((size_t)wave * G * kWave + lane) * TPW;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int q = 0; q < TPW; ++q) wres[g * TPW + q] = ap[(size_t)g * kWave * TPW + q];
    }
    int par = 0;                                           // parity the next tick reads
    for (int base = 0; base < n; base += kMfmaChunk) {
        const int cnt = n - base < kMfmaChunk ? n - base : kMfmaChunk;
        // ---- stage this chunk's audio rows (zeros for warm-up and for streams that do not run the model)
        for (int sl = wave; sl < NS; sl += kMfmaWaves) {
            const int sg = s_base + sl;
            const bool lv = livef[sl] != 0.f;
            float* row = xb + sl * nP;
            if (lv && mode == MODE_CHAIN) {
                const float* src = a.out + (size_t)sg * n + base;
                if (((n | base) & 3) == 0) load_block(row, src, cnt, lane);
                else for (int t = lane; t < cnt; t += kWave) row[t] = src[t];
            } else if (lv && mode == MODE_NN_ONLY) {
                for (int t = lane; t < cnt; t += kWave) row[t] = a.in[(size_t)(base + t) * I];
            } else {
                for (int t = lane; t < cnt; t += kWave) row[t] = 0.f;
            }
        }
        __syncthreads();
        // the input column of frame 0
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
        if (tid < NS) write_xin(par, 0);
        __syncthreads();

        const int ticks = cnt + NL;
        for (int tick = 0; tick < ticks; ++tick) {
            const int rd = par, wr = par ^ 1;
            if (tid < NS && tick + 1 < cnt) write_xin(wr, tick + 1);

            // ---- Dense(H,1) + skip + output gain of frame tick-NL: wave w reduces streams 4w..4w+3
            const int fd = tick - NL;
            constexpr int kDenseWave0 = NW == 8 ? 4 : 0;   // wave 0 already writes the input column
            if (fd >= 0 && wave >= kDenseWave0 && wave < kDenseWave0 + 4) {
                const int sl = (wave - kDenseWave0) * 4 + (lane >> 4), q = lane & 15;
                const float* hv = hT + ((size_t)(NL - 1) * 2 + rd) * H * NS;
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < H / 16; ++j) part = __builtin_fmaf(wdl[q + 16 * j], hv[(q + 16 * j) * NS + sl], part);
                const float y = row_sum16(part) + wdl[H];
                const float x = xb[sl * nP + fd] * a.in_gain;
                float o = a.input_skip ? x + y : y;
                o = o * a.out_gain;
                if (q == 0 && livef[sl] != 0.f) xb[sl * nP + fd] = o;
            }

            // ---- recurrent layers, layer l on frame tick-l
            for (int li = 0; li < NL; ++li) {
                int l = li + l_first;                      // the second wave of a SIMD starts half a stack later
                if (l >= NL) l -= NL;
                const MfmaLayer& L = d.L[l];
                const int f = tick - l;
                float* h_rd = hT + ((size_t)l * 2 + rd) * H * NS;
                float* h_wr = hT + ((size_t)l * 2 + wr) * H * NS;
                float* cl = cT + (size_t)l * H * NS;
                if (f < 0 || f >= cnt) {                   // idle tick of this layer: carry the state over
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) {
                        const int e = (wave * TPW + tl) * 64 + lane;
                        h_wr[e] = h_rd[e];
                    }
                    continue;
                }
                f32x4 acc[TPW];
                const f32x4* bias4 = reinterpret_cast<const f32x4*>(bl + (size_t)l * H * 4);
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) acc[tl] = bias4[4 * (wave * TPW + tl) + (lane >> 4)];
                if (l == 0) {                              // the model inputs: one k-step (x, PARAM1, PARAM2, 0)
                    const float b = xin[rd * 64 + lane];
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl)
                        acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in0[tl], b, acc[tl], 0, 0, 0);
                }
                if constexpr (RES) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        float b[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) b[j] = h_rd[256 * g + 64 * j + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int tl = 0; tl < TPW; ++tl) {
                                const int e = j * TPW + tl;
                                acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[g * TPW + e / 4][e % 4], b[j], acc[tl], 0, 0, 0);
                            }
                    }
                } else {
                    const int g_in = l == 0 ? 0 : H / 16, g_tot = g_in + H / 16;
                    const f32x4* ap = reinterpret_cast<const f32x4*>(W + L.w_big_off) + ((size_t)wave * g_tot * kWave + lane) * TPW;
                    gates_mfma<TPW>(acc, ap, hT + ((size_t)(l > 0 ? l - 1 : 0) * 2 + rd) * H * NS, h_rd, g_in, g_tot, lane);
                }
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) {
                    const int e = (wave * TPW + tl) * 64 + lane;          // unit 4T + (lane>>4), stream lane&15
                    float hn;
                    if (L.cell == 0) {
                        const float gi = sigmoid_pre(acc[tl].x), gf = sigmoid_pre(acc[tl].y);
                        const float gg = tanh_rat(acc[tl].z), go = sigmoid_pre(acc[tl].w);
                        const float cn = __builtin_fmaf(gf, cl[e], gi * gg);
                        cl[e] = cn;
                        hn = go * tanh_rat(cn);
                    } else {
                        const float gz = sigmoid_pre(acc[tl].x), gr = sigmoid_pre(acc[tl].y);
                        const float nn = tanh_exp_pre(__builtin_fmaf(gr, acc[tl].z, acc[tl].w));      // (GRU: see GruCell::step)
                        hn = __builtin_fmaf(gz, h_rd[e] - nn, nn);
                    }
                    h_wr[e] = hn;
                }
            }
            __syncthreads();
            par = wr;
        }
        // ---- results of this chunk back to HBM
        for (int sl = wave; sl < NS; sl += kMfmaWaves) {
            const int sg = s_base + sl;
            if (livef[sl] == 0.f || mode == MODE_WARMUP) continue;
            float* dst = mode == MODE_CHAIN ? a.out + (size_t)sg * n + base : a.out + base;
            const float* row = xb + sl * nP;
            if (((n | base) & 3) == 0 && mode == MODE_CHAIN) store_block(dst, row, cnt, lane);
            else for (int t = lane; t < cnt; t += kWave) dst[t] = row[t];
        }
        __syncthreads();
    }

    // ---- recurrent state and smoother memories back to HBM for the streams that ran
    for (int l = 0; l < NL; ++l) {
        const MfmaLayer& L = d.L[l];
        for (int i = tid; i < H * NS; i += kMfmaThreads) {
            const int u = i / NS, sn = i % NS, sg = s_base + sn;
            if (sg < (int)a.n_streams && u < Ht && livef[sn] != 0.f) {
                float* stp = a.nn + (size_t)sg * a.nn_stride + L.state_off;
                stp[u] = hT[((size_t)l * 2 + par) * H * NS + i];
                if (L.cell == 0) stp[Ht + u] = cT[(size_t)l * H * NS + i];
            }
        }
    }
    if (tid < NS && mine_live && mode == MODE_CHAIN) {
        StreamState& st = a.st[s_base + tid];
        st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
        st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
        st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
        st.pending = pending;
    }
}

// ---------------------------------------------------------------- host side
typedef void (*MfmaFn)(LaunchArgs, MfmaDesc);
static MfmaFn mfma_fn(int hidden, bool resident)
{
    switch (hidden) {
#define AIDAX_MFMA_CASE(HID) case HID: return resident ? k_mfma<HID / 4 / mfma_waves(HID), mfma_waves(HID), true> \
                                                       : k_mfma<HID / 4 / mfma_waves(HID), mfma_waves(HID), false>
    AIDAX_MFMA_CASE(16); AIDAX_MFMA_CASE(32); AIDAX_MFMA_CASE(48); AIDAX_MFMA_CASE(64);
    AIDAX_MFMA_CASE(80); AIDAX_MFMA_CASE(96); AIDAX_MFMA_CASE(112); AIDAX_MFMA_CASE(128);
#undef AIDAX_MFMA_CASE
    default:  return nullptr;
    }
}

size_t mfma_lds_bytes(const MfmaDesc& d, uint32_t n_frames) { return mfma_lds_floats(d.hidden, d.n_layers, (int)n_frames) * sizeof(float); }

hipError_t launch_mfma_kernel(const LaunchArgs& a, const MfmaDesc& d, hipStream_t stream)
{
    MfmaFn fn = mfma_fn(d.hidden, d.n_layers == 1);
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = mfma_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    hipLaunchKernelGGL(fn, dim3(groups), dim3(mfma_waves(d.hidden) * kWave), lds, stream, a, d);
    return hipGetLastError();
}

}  // namespace aidax
