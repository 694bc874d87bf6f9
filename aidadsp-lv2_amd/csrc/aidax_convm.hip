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
//   D                    lane holds 4 consecutive frames of one output channel: tanh, one ds_write_b128
//                        into the other plane.
//
// Each wave keeps four frame tiles in flight so a B fragment read serves four MFMAs. Per-layer input
// history ((ksize-1)*dilation frames per input channel) persists in HBM in the same layout as k_conv,
// so the two kernels are interchangeable mid-stream. The DSP chain runs in the packed k_chain launches
// (split form); this kernel is applyModel only.
#include "aidax_device.h"
#include "aidax_kernels.h"
#include "aidax_layout.h"

namespace aidax {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kConvmThreads = 256;
constexpr int kConvmTiles = 4;            // frame tiles per wave per pass (256 frames = 16 tiles = 4 waves x 4)

// frames per channel row: history (rounded to 4) + block (rounded to whole tiles), then up to 16 (mod 64)
__host__ __device__ inline int convm_plane_stride(int max_hist, int n_frames)
{
    const int f = ((max_hist + 3) & ~3) + ((n_frames + 15) & ~15);
    return ((f + 47) / 64) * 64 + 16;
}
__host__ __device__ inline size_t convm_lds_floats(const ConvDesc& d, int n_frames)
{
    return 2 * (size_t)d.channels * convm_plane_stride(d.max_hist, n_frames)   /* two activation planes     */
         + 2 * (size_t)d.max_k_steps * kWave                                  /* B fragments + A offsets   */
         + 16 + 16 + 4;                                                       /* bias, Dense weights + bias */
}

__global__ __launch_bounds__(kConvmThreads) void k_conv_mfma(LaunchArgs a, ConvDesc d)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = (int)a.n_frames;
    const int sg = blockIdx.x;
    const int mode = a.mode;
    if (n == 0) return;
    if (mode == MODE_CHAIN) {
        const uint32_t flags = a.ctl[sg].flags;
        if (!(flags & CTL_ENABLED) || !(flags & CTL_NET_ON)) return;          // :607-619, :631-632 (uniform per workgroup)
    }
    const int C = d.channels;
    const int Hb = (d.max_hist + 3) & ~3;                    // plane index of frame 0
    const int F = convm_plane_stride(d.max_hist, n);
    float* pa = smem;
    float* pb = pa + (size_t)C * F;
    float* wst = pb + (size_t)C * F;                          // [k_steps][64] B fragments of the current layer
    int* ofs = reinterpret_cast<int*>(wst + (size_t)d.max_k_steps * kWave);   // [k_steps][64] A plane offsets
    float* bsh = reinterpret_cast<float*>(ofs + (size_t)d.max_k_steps * kWave);   // [16]
    float* wdl = bsh + 16;                                    // [16] + bias

    const float* W = a.wpack;
    float* hist_base = a.nn + (size_t)sg * a.nn_stride;
    float* row = mode == MODE_CHAIN ? a.out + (size_t)sg * n : a.out;
    const int n16 = (n + 15) & ~15;

    // layer-0 input plane: [history | x * in_gain | zeros up to the tile boundary]; x stays in a register for the skip
    float xg = 0.f;
    {
        const ConvLayer& L0 = d.L[0];
        for (int i = tid; i < L0.hist * L0.in_ch; i += kConvmThreads)
            pa[(i / L0.hist) * F + Hb - L0.hist + (i % L0.hist)] = hist_base[L0.state_off + i];
        for (int t = tid; t < n16; t += kConvmThreads) {
            float v = 0.f;
            if (t < n) v = (mode == MODE_CHAIN ? row[t] : mode == MODE_NN_ONLY ? a.in[(size_t)t * a.input_size] : 0.f) * a.in_gain;
            pa[Hb + t] = v;
            xg = v;                                           // n <= 256: one frame per thread
        }
        if (tid < 17) wdl[tid] = tid < 16 ? (tid < d.L[d.n_layers - 1].out_ch ? W[d.wd_off + tid] : 0.f) : W[d.bd_off];
    }
    float* cur = pa;
    float* nxt = pb;
    const int ntiles = n16 / 16;
    for (int l = 0; l < d.n_layers; ++l) {
        const ConvLayer& L = d.L[l];
        const int Ci = L.in_ch, Co = L.out_ch, Hs = L.hist, K = L.ksize * Ci;
        __syncthreads();                                      // previous layer's readers of wst/ofs are done; planes written
        for (int i = tid; i < L.k_steps * kWave; i += kConvmThreads) {
            wst[i] = W[L.wf_off + i];
            const int k = 4 * (i >> 6) + ((i & 63) >> 4);
            const int kc = k < K ? k : 0;                     // padding k-steps carry zero weights: point them at valid data
            const int tap = kc / Ci, cin = kc - tap * Ci;
            ofs[i] = cin * F + Hb - (L.ksize - 1 - tap) * L.dilation + (i & 15);
        }
        if (tid < 16) bsh[tid] = tid < Co ? W[L.b_off + tid] : 0.f;
        if (l + 1 < d.n_layers) {                             // the next layer's history prefix
            const ConvLayer& N = d.L[l + 1];
            for (int i = tid; i < N.hist * N.in_ch; i += kConvmThreads)
                nxt[(i / N.hist) * F + Hb - N.hist + (i % N.hist)] = hist_base[N.state_off + i];
        }
        __syncthreads();
        const float bias = bsh[lane & 15];
        // tile j of this wave is frame tile wave + 4*j (n <= 256: at most kConvmTiles per wave)
        if (wave < ntiles) {
            f32x4 acc[kConvmTiles];
#pragma unroll
            for (int j = 0; j < kConvmTiles; ++j) acc[j] = f32x4{bias, bias, bias, bias};
            for (int kk = 0; kk < L.k_steps; ++kk) {
                const float b = wst[kk * kWave + lane];
                const float* ap = cur + ofs[kk * kWave + lane] + 16 * wave;
#pragma unroll
                for (int j = 0; j < kConvmTiles; ++j) {
                    const float av = ap[wave + 4 * j < ntiles ? 64 * j : 0];   // tiles past the block re-read the first
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[j], 0, 0, 0);
                }
            }
            const int co = lane & 15;
#pragma unroll
            for (int j = 0; j < kConvmTiles; ++j) {
                if (wave + 4 * j >= ntiles || co >= Co) continue;
                f32x4 v = acc[j];
                if (L.activation == 1) { v.x = tanh_rat(v.x); v.y = tanh_rat(v.y); v.z = tanh_rat(v.z); v.w = tanh_rat(v.w); }
                else if (L.activation == 2) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
                else if (L.activation == 3) { v.x = fast_sigmoid(v.x); v.y = fast_sigmoid(v.y); v.z = fast_sigmoid(v.z); v.w = fast_sigmoid(v.w); }
                *reinterpret_cast<f32x4*>(nxt + (size_t)co * F + Hb + 16 * (wave + 4 * j) + 4 * (lane >> 4)) = v;
            }
        }
        __syncthreads();
        // this layer's new history: the last Hs frames of [old history | this block's inputs]
        for (int i = tid; i < Hs * Ci; i += kConvmThreads)
            hist_base[L.state_off + i] = cur[(i / Hs) * F + Hb - Hs + n + (i % Hs)];
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    __syncthreads();
    // Dense(C,1) + skip / output gain (:171-181), one thread per frame
    if (mode != MODE_WARMUP && tid < n) {
        const int Cl = d.L[d.n_layers - 1].out_ch;
        float y = wdl[16];
        for (int o = 0; o < Cl; ++o) y = __builtin_fmaf(wdl[o], cur[(size_t)o * F + Hb + tid], y);
        float o2 = a.input_skip ? xg + y : y;
        row[tid] = o2 * a.out_gain;
    }
}

size_t convm_lds_bytes(const ConvDesc& d, uint32_t n_frames) { return convm_lds_floats(d, (int)n_frames) * sizeof(float); }

hipError_t launch_conv_mfma_kernel(const LaunchArgs& a, const ConvDesc& d, hipStream_t stream)
{
    const size_t lds = convm_lds_bytes(d, a.n_frames);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_conv_mfma, dim3(a.n_streams), dim3(kConvmThreads), lds, stream, a, d);
    return hipGetLastError();
}

}  // namespace aidax
