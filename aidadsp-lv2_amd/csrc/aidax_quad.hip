// aidax_quad.hip — k_quad: the recurrent cell of the reference's table models for the many-streams regime,
// FOUR streams per workgroup on v_mfma_f32_4x4x1_16b_f32.
//
// That instruction multiplies sixteen independent 4x1 by 1x4 blocks: block b's D[i][j] += A[i] * B[j].
// Here block b is hidden unit b of the wave (16 units per wave), i the unit's four gate rows (LSTM i,f,g,o;
// GRU z, r, recurrent / input half of the candidate) and j one of four streams:
//   A  lane l holds the weight of gate row (unit l/4, row l%4) for contraction column k — ONE REGISTER PER
//      COLUMN, resident for the whole launch like the one-wave kernels (H+3 registers: 35 for LSTM-32);
//   B  lane l supplies h_{stream l%4}[k]: a float4 of the stream's h row in LDS feeds four MFMAs;
//   D  lane l ends up with all four gates of unit l/4 for stream l%4: the cell update is lane-local, c (and
//      the GRU's previous h) never leave their register, and all 64 lanes do useful activation work.
// One MFMA (8.7 cycles measured, scratch/umfma4.hip) replaces 4 streams x 64 rows = 256 scalar FMAs, i.e.
// the 72-FMA inner loop of LSTM-32 becomes 35 MFMAs for four streams. ceil(H/16) waves per workgroup
// share the h rows through a 16-frame LDS ring (one LDS-only barrier per frame when there is more than one
// wave); wave 0 does the Dense for 8 frames at a time, one lane per (frame, stream).
// The DSP chain runs in the packed k_chain launches around this kernel (split form).
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQuadStreams = 4;
constexpr int kQuadRing = 16;             // frames of h history (Dense runs 8 frames behind at most); LDS per workgroup decides residency
constexpr int kQuadSub = 8;               // frames between Dense passes

__host__ __device__ constexpr int quad_waves(int hidden) { return (hidden + 15) / 16; }
__host__ __device__ constexpr int quad_row_stride(int hidden) { return hidden + 4; }     // 4 streams' float4 reads on disjoint banks
__host__ __device__ inline size_t quad_lds_floats(int hidden, int n_frames)
{
    return (size_t)kQuadStreams * ((n_frames + 3) & ~3)                       /* xb: the four audio rows   */
         + (size_t)kQuadRing * kQuadStreams * quad_row_stride(hidden)         /* h ring                    */
         + (size_t)((hidden + 1 + 3) & ~3);                                   /* Dense weights + bias      */
}

__device__ __forceinline__ void quad_barrier(bool multi_wave)
{
    if (multi_wave) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __builtin_amdgcn_wave_barrier();
}

template <int CELL, int H>
__global__ __launch_bounds__(quad_waves(H) * kWave) void k_quad(LaunchArgs a, QuadDesc qd)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = quad_waves(H), HS = quad_row_stride(H), KR = H + kMaxInputs;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 3;                                   // stream of this lane's accumulators
    const int u = 16 * wave + (lane >> 2);                    // unit of this lane's accumulators (and weight row)
    const bool uok = u < H;
    const int n = (int)a.n_frames;
    const int nP = (n + 3) & ~3;
    const int mode = a.mode;
    const int I = a.input_size;
    const int sg = blockIdx.x * kQuadStreams + j;
    const bool valid = sg < (int)a.n_streams;
    const int sc = valid ? sg : (int)a.n_streams - 1;

    float* xb  = smem;                                        // [4][nP]
    float* hh  = xb + kQuadStreams * nP;                      // [ring][4][HS]
    float* wdl = hh + kQuadRing * kQuadStreams * HS;          // Dense weights, bias at [H]

    const StreamCtl& ctl = a.ctl[sc];
    StreamState& st = a.st[sc];
    bool live = valid && n != 0;
    if (mode == MODE_CHAIN) live = live && (ctl.flags & CTL_ENABLED) && (ctl.flags & CTL_NET_ON);    // :607-619, :631-632
    else if (mode == MODE_NN_ONLY) live = live && sg == 0;
    if (__builtin_amdgcn_ballot_w64(live) == 0) return;       // same four streams in every wave: uniform exit

    // ---- weights of this lane's gate row, resident for the launch
    const float* W = a.wpack;
    float wr[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) wr[k] = W[((size_t)wave * KR + k) * kWave + lane];
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(W + qd.bias_off + (size_t)u * 4);
    for (int i = tid; i < H + 1; i += NW * kWave) wdl[i] = W[qd.dense_off + i];

    // ---- audio rows of the four streams (in place on `out`; zeros for warm-up and rows that do not run)
    float* rows = a.out;
    for (int g = 0; g < kQuadStreams; ++g) {
        const bool row_live = __builtin_amdgcn_readlane((int)live, g) != 0;
        float* dst = xb + g * nP;
        const int s2 = blockIdx.x * kQuadStreams + g;
        if (row_live && mode == MODE_CHAIN) {
            for (int t = tid; t < n; t += NW * kWave) dst[t] = rows[(size_t)s2 * n + t];
        } else if (row_live && mode == MODE_NN_ONLY) {
            for (int t = tid; t < n; t += NW * kWave) dst[t] = a.in[(size_t)t * I];
        } else {
            for (int t = tid; t < n; t += NW * kWave) dst[t] = 0.f;
        }
    }

    // ---- recurrent state and PARAM smoothers
    float* nnst = a.nn + (size_t)sc * a.nn_stride;
    float hcur = (valid && uok) ? nnst[u] : 0.f;
    float c = (CELL == 0 && valid && uok) ? nnst[H + u] : 0.f;
    if (uok) hh[((kQuadRing - 1) * kQuadStreams + j) * HS + u] = hcur;
    float p_mem[2] = { st.p_mem[0], st.p_mem[1] };
    float p_tgt[2] = { st.p_tgt[0], st.p_tgt[1] };
    float p_step[2] = { st.p_step[0], st.p_step[1] };
    uint32_t pending = st.pending;
    if (mode == MODE_CHAIN && live) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                        // LinearValueSmoother::setTargetValue (:209-216)
            const float nt = ctl.p_target[i];
            if (__builtin_fabsf(p_tgt[i] - nt) >= FLT_EPSILON) {
                p_tgt[i] = nt;
                p_step[i] = (p_tgt[i] - p_mem[i]) / ctl.p_den;
            }
        }
        if (pending & PEND_PARAM_FIRST) {                    // paramFirstRun (:636-640)
            pending &= ~PEND_PARAM_FIRST;
            p_mem[0] = p_tgt[0];
            p_mem[1] = p_tgt[1];
        }
    }
    __syncthreads();

    const float in_gain = a.in_gain;
    const float* xrow = xb + j * nP;
    for (int base = 0; base < n; base += kQuadSub) {
        const int cnt = n - base < kQuadSub ? n - base : kQuadSub;
        float xnext = xrow[base];
        for (int t = base; t < base + cnt; ++t) {
            const float x = xnext * in_gain;
            xnext = xrow[t + 1 < n ? t + 1 : t];
            f32x4 acc0 = bias4, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[H], x, acc0, 0, 0, 0);
            if (I >= 2) {                                     // PARAM1 / PARAM2 (:195-231)
                float q1, q2 = 0.f;
                if (mode == MODE_CHAIN) {
                    q1 = lin_next(p_mem[0], p_tgt[0], p_step[0]);
                    if (I >= 3) q2 = lin_next(p_mem[1], p_tgt[1], p_step[1]);
                } else if (mode == MODE_WARMUP) {             // constant params over the zero pre-buffer (:1077-1078)
                    q1 = p_mem[0];
                    q2 = I >= 3 ? p_mem[1] : 0.f;
                } else {
                    q1 = a.in[(size_t)t * I + 1];
                    q2 = I >= 3 ? a.in[(size_t)t * I + 2] : 0.f;
                }
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[H + 1], q1, acc1, 0, 0, 0);
                if (I >= 3) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[H + 2], q2, acc0, 0, 0, 0);
            }
            const f32x4* hv = reinterpret_cast<const f32x4*>(hh + (((t + kQuadRing - 1) & (kQuadRing - 1)) * kQuadStreams + j) * HS);
#pragma unroll
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const f32x4 h4 = hv[k4];
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4], h4.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4 + 1], h4.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4 + 2], h4.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[4 * k4 + 3], h4.w, acc1, 0, 0, 0);
            }
            const f32x4 g = acc0 + acc1;
            if constexpr (CELL == 0) {
                const float gi = fast_sigmoid(g.x), gf = fast_sigmoid(g.y);
                const float gg = tanh_rat(g.z), go = fast_sigmoid(g.w);
                c = __builtin_fmaf(gf, c, gi * gg);
                hcur = go * tanh_rat(c);
            } else {
                const float gz = fast_sigmoid(g.x), gr = fast_sigmoid(g.y);
                const float nn = tanh_exp(__builtin_fmaf(gr, g.z, g.w));      // (GRU: see GruCell::step)
                hcur = __builtin_fmaf(gz, hcur - nn, nn);
            }
            if (uok) hh[((t & (kQuadRing - 1)) * kQuadStreams + j) * HS + u] = hcur;
            quad_barrier(NW > 1);
        }
        // Dense(H,1) + skip + output gain for these frames: wave 0, one lane per (frame, stream)
        if (wave == 0) {
            const int f = lane >> 2;
            const int tf = base + (f < cnt ? f : cnt - 1);
            const f32x4* hrow = reinterpret_cast<const f32x4*>(hh + ((tf & (kQuadRing - 1)) * kQuadStreams + j) * HS);
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wdl);
            float y = wdl[H];
#pragma unroll 2
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const f32x4 w = w4[k4];
                const f32x4 hq = hrow[k4];
                y = __builtin_fmaf(w.x, hq.x, y);
                y = __builtin_fmaf(w.y, hq.y, y);
                y = __builtin_fmaf(w.z, hq.z, y);
                y = __builtin_fmaf(w.w, hq.w, y);
            }
            const float xg = xrow[tf] * in_gain;
            float o = a.input_skip ? xg + y : y;
            o = o * a.out_gain;
            if (f < cnt && live) xb[j * nP + tf] = o;
        }
    }
    __syncthreads();

    // ---- results and state back to HBM
    if (mode != MODE_WARMUP) {
        for (int g = 0; g < kQuadStreams; ++g) {
            if (__builtin_amdgcn_readlane((int)live, g) == 0) continue;
            const int s2 = blockIdx.x * kQuadStreams + g;
            float* dst = mode == MODE_CHAIN ? rows + (size_t)s2 * n : rows;
            for (int t = tid; t < n; t += NW * kWave) dst[t] = xb[g * nP + t];
        }
    }
    if (live && uok) {
        nnst[u] = hcur;
        if (CELL == 0) nnst[H + u] = c;
    }
    if (live && mode == MODE_CHAIN && tid < kQuadStreams) {   // lanes 0..3 of wave 0: one per stream
        st.p_mem[0] = p_mem[0]; st.p_mem[1] = p_mem[1];
        st.p_tgt[0] = p_tgt[0]; st.p_tgt[1] = p_tgt[1];
        st.p_step[0] = p_step[0]; st.p_step[1] = p_step[1];
        st.pending = pending;
    }
}

// ---------------------------------------------------------------- host side
typedef void (*QuadFn)(LaunchArgs, QuadDesc);
struct QuadEntry { int cell, hidden; QuadFn fn; };
#define AIDAX_QUAD(H) { 0, H, k_quad<0, H> }, { 1, H, k_quad<1, H> }
static const QuadEntry kQuadTable[] = {
    AIDAX_QUAD(8), AIDAX_QUAD(12), AIDAX_QUAD(16), AIDAX_QUAD(20), AIDAX_QUAD(24),
    AIDAX_QUAD(32), AIDAX_QUAD(40), AIDAX_QUAD(64), AIDAX_QUAD(80),
};

size_t quad_lds_bytes(int hidden, uint32_t n_frames) { return quad_lds_floats(hidden, (int)n_frames) * sizeof(float); }

hipError_t launch_quad_kernel(int cell, int hidden, const LaunchArgs& a, const QuadDesc& qd, hipStream_t stream)
{
    QuadFn fn = nullptr;
    for (const auto& e : kQuadTable)
        if (e.cell == cell && e.hidden == hidden) fn = e.fn;
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = quad_lds_bytes(hidden, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t groups = (a.n_streams + kQuadStreams - 1) / kQuadStreams;
    hipLaunchKernelGGL(fn, dim3(groups), dim3(quad_waves(hidden) * kWave), lds, stream, a, qd);
    return hipGetLastError();
}

}  // namespace aidax
