// aidax_stack.hip — the VALU kernels for the architectures the reference itself cannot load
// (BASELINE configs #4 and #5; parity pinned only by the torch fixtures of
// tests/golden/make_golden.py). The pool prefers their matrix-core counterparts (k_mfma in
// aidax_mfma.hip, k_conv_mfma in aidax_convm.hip); these serve widths / block lengths those do
// not take and AIDAX_KERNEL=valu, and the tests run both forms against the same oracle:
//
//   k_stack   stacked LSTM/GRU layers (e.g. LSTM-96 x2, 437 KiB of weights) -> Dense(H,1).
//             The weights do not fit one wave's registers, so a 256-thread workgroup
//             carries kStackStreams streams: each thread owns gate rows (k-major weights,
//             coalesced, L2-resident) and reuses every weight it loads across the
//             workgroup's streams; h/c live in LDS; gate rows and unit updates are
//             separated by workgroup barriers.
//   k_conv    causal dilated conv1d stack (k taps, dilation 2^l, tanh) -> Dense(C,1).
//             Feed-forward, so a 256-frame block is computed time-parallel: one thread
//             per frame, layer by layer through two LDS activation planes; each layer's
//             (k-1)*dilation frames of input history persist in HBM between blocks.
//
// Both run the same pre/post systolic biquad passes, ramps and control semantics as
// aidax_kernels.hip (one wave handles the chain of two streams resp. the whole stream).
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

// PARAM1/2 ramps of one block (:634-640 + LinearValueSmoother::next), written to pq[t][2]
__device__ __forceinline__ void param_ramps(const StreamCtl& ctl, StreamState& st, ChainCtx& c, float* pq, int n, int I, int lane)
{
    float p_mem[2] = { st.p_mem[0], st.p_mem[1] };
    float p_tgt[2] = { st.p_tgt[0], st.p_tgt[1] };
    float p_step[2] = { st.p_step[0], st.p_step[1] };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float nt = ctl.p_target[i];
        if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
            p_tgt[i] = nt;
            p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
        }
    }
    if (c.pending & PEND_PARAM_FIRST) {
        c.pending &= ~PEND_PARAM_FIRST;
        p_mem[0] = p_tgt[0];
        p_mem[1] = p_tgt[1];
    }
    if (I >= 2) {
        for (int t = 0; t < n; ++t) {
            const float q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
            const float q2 = I >= 3 ? lin_next(p_mem[1], p_tgt[1], p_step[1]) : 0.f;
            if (lane == 0) { pq[2 * t] = q1; pq[2 * t + 1] = q2; }
        }
    }
    if (lane == 0) {
        st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
        st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
        st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
    }
}

// ====================================================================== k_stack
constexpr int kStackThreads = 256;
constexpr int kStackChunks = 2;          // row chunks per thread: rows <= 512 (hidden <= 128 for LSTM)

__host__ __device__ inline size_t stack_lds_floats(const StackDesc& d, int n_frames)
{
    const size_t nP = (size_t)((n_frames + 3) & ~3);
    const size_t S = kStackStreams;
    return S * nP                      /* xbuf  */
         + S * nP * 2                  /* pbuf  */
         + S * 4                       /* vin   */
         + (size_t)d.n_layers * S * d.max_hidden * 2   /* hbuf, cbuf */
         + S * d.max_rows              /* z     */
         + S * d.max_hidden            /* z2    */
         + S;                          /* live flags */
}

// acc[j][s] += sum_k Wt[k][r_j] * v[s][k] for k in [0,K): v rows are `vstride` floats apart in LDS
template <bool VEC4>
__device__ __forceinline__ void rows_accumulate(float (&acc)[kStackChunks][kStackStreams], const float* __restrict__ wt,
                                                int R, int K, const float* v, int vstride, int tid)
{
    // rows past R are clamped to a valid row instead of predicated: their sums are never stored, and a
    // guarded load costs a branch + exec save/restore per weight
    const int r0 = tid < R ? tid : R - 1;
    const int r1 = tid + kStackThreads < R ? tid + kStackThreads : R - 1;
    if constexpr (VEC4) {
#pragma unroll 4
        for (int k = 0; k < K; k += 4) {
            float w0[4], w1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w0[i] = wt[(size_t)(k + i) * R + r0];
                w1[i] = wt[(size_t)(k + i) * R + r1];
            }
#pragma unroll
            for (int s = 0; s < kStackStreams; ++s) {
                const float4 x = *reinterpret_cast<const float4*>(v + s * vstride + k);    // LDS broadcast
                acc[0][s] = __builtin_fmaf(w0[0], x.x, acc[0][s]);
                acc[0][s] = __builtin_fmaf(w0[1], x.y, acc[0][s]);
                acc[0][s] = __builtin_fmaf(w0[2], x.z, acc[0][s]);
                acc[0][s] = __builtin_fmaf(w0[3], x.w, acc[0][s]);
                acc[1][s] = __builtin_fmaf(w1[0], x.x, acc[1][s]);
                acc[1][s] = __builtin_fmaf(w1[1], x.y, acc[1][s]);
                acc[1][s] = __builtin_fmaf(w1[2], x.z, acc[1][s]);
                acc[1][s] = __builtin_fmaf(w1[3], x.w, acc[1][s]);
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const float w0 = wt[(size_t)k * R + r0];
            const float w1 = wt[(size_t)k * R + r1];
#pragma unroll
            for (int s = 0; s < kStackStreams; ++s) {
                const float x = v[s * vstride + k];
                acc[0][s] = __builtin_fmaf(w0, x, acc[0][s]);
                acc[1][s] = __builtin_fmaf(w1, x, acc[1][s]);
            }
        }
    }
}

__global__ __launch_bounds__(kStackThreads) void k_stack(LaunchArgs a, StackDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int nP = (n + 3) & ~3;
    constexpr int S = kStackStreams;
    const int s_base = blockIdx.x * S;
    const int HM = d.max_hidden;

    float* xbuf = smem;                              // [S][nP]
    float* pbuf = xbuf + S * nP;                     // [S][nP][2]
    float* vin  = pbuf + S * nP * 2;                 // [S][4]
    float* hbuf = vin + S * 4;                       // [L][S][HM]
    float* cbuf = hbuf + d.n_layers * S * HM;        // [L][S][HM]
    float* zb   = cbuf + d.n_layers * S * HM;        // [S][max_rows]
    float* z2   = zb + S * d.max_rows;               // [S][HM]
    float* livef = z2 + S * HM;                      // [S]: 1 when the NN result is committed

    const bool bare = a.mode != MODE_CHAIN;
    const int I = a.input_size;

    // ---- phase 1: chain prologue / pre pass; each wave takes two streams
    ChainCtx ctx[2];
    for (int j = 0; j < 2; ++j) {
        const int sl = wave * 2 + j;
        const int sg = s_base + sl;
        ctx[j].live = false;
        if (sg >= (int)a.n_streams) { if (lane == 0) livef[sl] = 0.f; continue; }
        if (bare) {                                   // warm-up: zeros in, constant params (:1077-1078)
            StreamState& st = a.st[sg];
            for (int t = lane; t < n; t += kWave) {
                xbuf[sl * nP + t] = a.mode == MODE_NN_ONLY ? a.in[(size_t)t * I] : 0.f;
                pbuf[(sl * nP + t) * 2 + 0] = a.mode == MODE_NN_ONLY ? (I >= 2 ? a.in[(size_t)t * I + 1] : 0.f) : st.p_mem[0];
                pbuf[(sl * nP + t) * 2 + 1] = a.mode == MODE_NN_ONLY ? (I >= 3 ? a.in[(size_t)t * I + 2] : 0.f) : st.p_mem[1];
            }
            if (lane == 0) livef[sl] = 1.f;
            continue;
        }
        const StreamCtl& ctl = a.ctl[sg];
        StreamState& st = a.st[sg];
        ctx[j] = chain_prologue(ctl, st, a.in + (size_t)sg * n, a.out + (size_t)sg * n, xbuf + sl * nP, n, lane);
        const bool net = ctx[j].live && (ctx[j].flags & CTL_NET_ON);
        if (net) param_ramps(ctl, st, ctx[j], pbuf + sl * nP * 2, n, I, lane);
        if (lane == 0) livef[sl] = net ? 1.f : 0.f;
    }
    // recurrent state -> LDS
    for (int l = 0; l < d.n_layers; ++l) {
        const StackLayer& L = d.L[l];
        for (int i = tid; i < S * L.hidden; i += kStackThreads) {
            const int sl = i / L.hidden, u = i % L.hidden, sg = s_base + sl;
            const float* stp = a.nn + (size_t)(sg < (int)a.n_streams ? sg : 0) * a.nn_stride + L.state_off;
            hbuf[(l * S + sl) * HM + u] = sg < (int)a.n_streams ? stp[u] : 0.f;
            cbuf[(l * S + sl) * HM + u] = (sg < (int)a.n_streams && L.cell == 0) ? stp[L.hidden + u] : 0.f;
        }
    }
    __syncthreads();

    // ---- phase 2: the recurrent stack, all 256 threads on the workgroup's streams
    const float* W = a.wpack;
    for (int t = 0; t < n; ++t) {
        if (tid < S) {
            const float x = xbuf[tid * nP + t] * a.in_gain;
            vin[tid * 4 + 0] = x;
            vin[tid * 4 + 1] = pbuf[(tid * nP + t) * 2 + 0];
            vin[tid * 4 + 2] = pbuf[(tid * nP + t) * 2 + 1];
            vin[tid * 4 + 3] = 0.f;
        }
        __syncthreads();
        for (int l = 0; l < d.n_layers; ++l) {
            const StackLayer& L = d.L[l];
            const int R = L.rows, H = L.hidden;
            float ax[kStackChunks][S], ah[kStackChunks][S];
#pragma unroll
            for (int j = 0; j < kStackChunks; ++j)
#pragma unroll
                for (int s = 0; s < S; ++s) { ax[j][s] = 0.f; ah[j][s] = 0.f; }
            // input part: W^T x  (layer 0: [x,p1,p2]; deeper: h of the layer below, this frame)
            if (l == 0) rows_accumulate<false>(ax, W + L.w_off, R, L.in_size, vin, 4, tid);
            else rows_accumulate<true>(ax, W + L.w_off, R, L.in_size, hbuf + (l - 1) * S * HM, HM, tid);
            // recurrent part: U^T h(t-1)
            rows_accumulate<true>(ah, W + L.w_off + (size_t)L.in_size * R, R, H, hbuf + l * S * HM, HM, tid);
#pragma unroll
            for (int j = 0; j < kStackChunks; ++j) {
                const int r = tid + j * kStackThreads;
                if (r < R) {
                    const float b = W[L.b_off + r];
                    const bool gru_n = L.cell == 1 && r >= 2 * H;
                    const float b2 = gru_n ? W[L.b2_off + (r - 2 * H)] : 0.f;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if (gru_n) { zb[s * d.max_rows + r] = ax[j][s] + b; z2[s * HM + (r - 2 * H)] = ah[j][s] + b2; }
                        else zb[s * d.max_rows + r] = (ah[j][s] + b) + ax[j][s];
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < S * H; i += kStackThreads) {
                const int sl = i / H, u = i % H;
                const float* z = zb + sl * d.max_rows;
                float hn;
                if (L.cell == 0) {
                    const float gi = fast_sigmoid(z[u]), gf = fast_sigmoid(z[H + u]);
                    const float gg = tanh_rat(z[2 * H + u]), go = fast_sigmoid(z[3 * H + u]);
                    const float cn = __builtin_fmaf(gf, cbuf[(l * S + sl) * HM + u], gi * gg);
                    cbuf[(l * S + sl) * HM + u] = cn;
                    hn = go * tanh_rat(cn);
                } else {
                    const float gz = fast_sigmoid(z[u]), gr = fast_sigmoid(z[H + u]);
                    const float nn = tanh_rat(__builtin_fmaf(gr, z2[sl * HM + u], z[2 * H + u]));
                    const float ho = hbuf[(l * S + sl) * HM + u];
                    hn = __builtin_fmaf(gz, ho - nn, nn);
                }
                hbuf[(l * S + sl) * HM + u] = hn;
            }
            __syncthreads();
        }
        {   // Dense(H,1) + skip/out gain (:171-181): wave w reduces streams 2w and 2w+1
            const int Hl = d.L[d.n_layers - 1].hidden;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int sl = wave * 2 + j;
                const float* hv = hbuf + ((d.n_layers - 1) * S + sl) * HM;
                float part = 0.f;
                for (int u = lane; u < Hl; u += kWave) part = __builtin_fmaf(W[d.wd_off + u], hv[u], part);
                const float y = wave_sum(part) + W[d.bd_off];
                const float x = vin[sl * 4];
                float o = a.input_skip ? x + y : y;
                o = o * a.out_gain;
                if (lane == 0 && livef[sl] != 0.f) {
                    if (a.mode == MODE_NN_ONLY) { if (s_base + sl == 0) a.out[t] = o; }
                    else xbuf[sl * nP + t] = o;
                }
            }
        }
        __syncthreads();                                         // vin is rewritten at the top of the next frame
    }
    __syncthreads();
    // recurrent state back to HBM for committed streams
    for (int l = 0; l < d.n_layers; ++l) {
        const StackLayer& L = d.L[l];
        for (int i = tid; i < S * L.hidden; i += kStackThreads) {
            const int sl = i / L.hidden, u = i % L.hidden, sg = s_base + sl;
            if (sg < (int)a.n_streams && livef[sl] != 0.f) {
                float* stp = a.nn + (size_t)sg * a.nn_stride + L.state_off;
                stp[u] = hbuf[(l * S + sl) * HM + u];
                if (L.cell == 0) stp[L.hidden + u] = cbuf[(l * S + sl) * HM + u];
            }
        }
    }
    if (bare) return;

    // ---- phase 3: post pass + store, two streams per wave
    for (int j = 0; j < 2; ++j) {
        const int sl = wave * 2 + j;
        const int sg = s_base + sl;
        if (sg >= (int)a.n_streams || !ctx[j].live) continue;
        chain_epilogue(a.ctl[sg], a.st[sg], ctx[j], a.out + (size_t)sg * n, xbuf + sl * nP, n, lane);
    }
}

// ====================================================================== k_conv
// LDS: two activation planes, channel-major [C][hist_max + n (+pad)] so that the 64 lanes of a wave
// (consecutive frames) read consecutive LDS words, + the stream's audio block.
constexpr int kConvCo = 16;                      // output channels are padded to 16 accumulators
constexpr int kConvWStage = 8 * 16 * kConvCo + kConvCo;   // one layer's kernel [k<=8][in<=16][16] + bias, staged in LDS
__host__ __device__ inline int conv_plane_stride(int max_hist, int n_frames) { return ((max_hist + n_frames + 3) & ~3) + 1; }
__host__ __device__ inline size_t conv_lds_floats(const ConvDesc& d, int n_frames)
{
    const size_t nP = (size_t)((n_frames + 3) & ~3);
    return nP + nP * 2 + 2 * (size_t)conv_plane_stride(d.max_hist, n_frames) * d.channels + 64 + kConvWStage;
}

__global__ __launch_bounds__(kStackThreads) void k_conv(LaunchArgs a, ConvDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int nP = (n + 3) & ~3;
    const int sg = blockIdx.x;
    const int C = d.channels;
    const int F = conv_plane_stride(d.max_hist, n);          // frames per channel row
    const size_t plane = (size_t)F * C;

    float* buf = smem;                      // audio block, chain in place
    float* pq = buf + nP;                   // PARAM ramps (kept for interface symmetry; conv models have I = 1..3)
    float* pa = pq + nP * 2;                // activation plane A
    float* pb = pa + plane;                 // activation plane B
    float* shared_flag = pb + plane;
    float* wst = shared_flag + 64;          // staged weights of the current layer, [k][in][16] then bias[16]

    const bool bare = a.mode != MODE_CHAIN;
    ChainCtx ctx;
    ctx.live = false;
    bool net = true;
    if (wave == 0) {
        if (bare) {
            for (int t = lane; t < n; t += kWave) buf[t] = a.mode == MODE_NN_ONLY ? a.in[(size_t)t * a.input_size] : 0.f;
        } else {
            ctx = chain_prologue(a.ctl[sg], a.st[sg], a.in + (size_t)sg * n, a.out + (size_t)sg * n, buf, n, lane);
            net = ctx.live && (ctx.flags & CTL_NET_ON);
            if (net) {
                uint32_t pend = ctx.pending;
                if (lane == 0) pend = param_targets(a.ctl[sg], a.st[sg], pend);
                ctx.pending = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
            }
        }
        if (lane == 0) { shared_flag[0] = net ? 1.f : 0.f; shared_flag[1] = (bare || ctx.live) ? 1.f : 0.f; }
    }
    __syncthreads();
    const bool run_net = shared_flag[0] != 0.f;
    const bool live = shared_flag[1] != 0.f;
    if (!live) return;

    if (run_net) {
        const float* W = a.wpack;
        float* hist_base = a.nn + (size_t)sg * a.nn_stride;
        float* cur = pa;
        float* nxt = pb;
        // layer-0 input plane: [hist0 | x*in_gain], one channel per input (audio only; params unsupported here)
        {
            const ConvLayer& L0 = d.L[0];
            for (int i = tid; i < L0.hist * L0.in_ch; i += kStackThreads)          // history kept as [ch][hist]
                cur[(i / L0.hist) * F + (i % L0.hist)] = hist_base[L0.state_off + i];
            for (int t = tid; t < n; t += kStackThreads) cur[L0.hist + t] = buf[t] * a.in_gain;
        }
        __syncthreads();
        for (int l = 0; l < d.n_layers; ++l) {
            const ConvLayer& L = d.L[l];
            const int Ci = L.in_ch, Co = L.out_ch, Hs = L.hist;
            const int next_hist = l + 1 < d.n_layers ? d.L[l + 1].hist : 0;
            // next layer's history prefix
            if (l + 1 < d.n_layers) {
                const ConvLayer& N = d.L[l + 1];
                for (int i = tid; i < N.hist * N.in_ch; i += kStackThreads)
                    nxt[(i / N.hist) * F + (i % N.hist)] = hist_base[N.state_off + i];
            }
            // stage this layer's kernel + bias in LDS (zero-padded to 16 outputs): every thread reads the
            // same weights, so they come back as LDS broadcasts instead of one global load per FMA
            for (int i = tid; i < L.ksize * Ci * kConvCo; i += kStackThreads) {
                const int o = i % kConvCo, ki = i / kConvCo;
                wst[i] = o < Co ? W[L.w_off + (size_t)ki * Co + o] : 0.f;
            }
            if (tid < kConvCo) wst[L.ksize * Ci * kConvCo + tid] = tid < Co ? W[L.b_off + tid] : 0.f;
            __syncthreads();
            const float4* bias4 = reinterpret_cast<const float4*>(wst + L.ksize * Ci * kConvCo);
            for (int t = tid; t < n; t += kStackThreads) {
                float acc[kConvCo];
#pragma unroll
                for (int q = 0; q < kConvCo / 4; ++q) {
                    const float4 b = bias4[q];
                    acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
                }
                for (int k = 0; k < L.ksize; ++k) {
                    const float* xs = cur + (Hs + t - (L.ksize - 1 - k) * L.dilation);
                    const float4* wk = reinterpret_cast<const float4*>(wst + (size_t)k * Ci * kConvCo);
                    for (int i = 0; i < Ci; ++i) {
                        const float xi = xs[(size_t)i * F];
#pragma unroll
                        for (int q = 0; q < kConvCo / 4; ++q) {
                            const float4 w = wk[i * (kConvCo / 4) + q];
                            acc[4 * q]     = __builtin_fmaf(w.x, xi, acc[4 * q]);
                            acc[4 * q + 1] = __builtin_fmaf(w.y, xi, acc[4 * q + 1]);
                            acc[4 * q + 2] = __builtin_fmaf(w.z, xi, acc[4 * q + 2]);
                            acc[4 * q + 3] = __builtin_fmaf(w.w, xi, acc[4 * q + 3]);
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < kConvCo; ++o) {
                    if (o < Co) {
                        float v = acc[o];
                        if (L.activation == 1) v = tanh_exp(v);
                        else if (L.activation == 2) v = v > 0.f ? v : 0.f;
                        else if (L.activation == 3) v = fast_sigmoid(v);
                        nxt[(size_t)o * F + next_hist + t] = v;
                    }
                }
            }
            __syncthreads();
            // this layer's new history: the last Hs frames of [old history | this block's inputs]
            for (int i = tid; i < Hs * Ci; i += kStackThreads)
                hist_base[L.state_off + i] = cur[(i / Hs) * F + n + (i % Hs)];
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
        // Dense(C,1) + skip/out gain
        const int Cl = d.L[d.n_layers - 1].out_ch;
        for (int t = tid; t < n; t += kStackThreads) {
            float y = W[d.bd_off];
            for (int o = 0; o < Cl; ++o) y = __builtin_fmaf(W[d.wd_off + o], cur[(size_t)o * F + t], y);
            const float x = buf[t] * a.in_gain;
            float o2 = a.input_skip ? x + y : y;
            o2 = o2 * a.out_gain;
            if (a.mode == MODE_NN_ONLY) { if (sg == 0) a.out[t] = o2; }
            else buf[t] = o2;
        }
        __syncthreads();
    }
    if (bare) return;
    if (wave == 0) chain_epilogue(a.ctl[sg], a.st[sg], ctx, a.out + (size_t)sg * n, buf, n, lane);
}

// ---------------------------------------------------------------- host side
size_t stack_lds_bytes(const StackDesc& d, uint32_t n_frames) { return stack_lds_floats(d, (int)n_frames) * sizeof(float); }
size_t conv_lds_bytes(const ConvDesc& d, uint32_t n_frames) { return conv_lds_floats(d, (int)n_frames) * sizeof(float); }

hipError_t launch_stack_kernel(const LaunchArgs& a, const StackDesc& d, hipStream_t stream)
{
    const size_t lds = stack_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stack), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kStackStreams - 1) / kStackStreams;
    hipLaunchKernelGGL(k_stack, dim3(groups), dim3(kStackThreads), lds, stream, a, d);
    return hipGetLastError();
}

hipError_t launch_conv_kernel(const LaunchArgs& a, const ConvDesc& d, hipStream_t stream)
{
    const size_t lds = conv_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_conv, dim3(a.n_streams), dim3(kStackThreads), lds, stream, a, d);
    return hipGetLastError();
}

}  // namespace aidax
