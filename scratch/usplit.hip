// Would splitting one stream's LSTM-32 cell over TWO waves (16 units each, S=4 lanes per unit) beat the
// one-wave cell (32 units, S=2)? Models the per-frame critical loop: h broadcast reads, FMAs against
// register weights, permlane exchange, activations, h publish, and the cross-wave hand-over
// (LDS flag spin or s_barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float sig(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.442695f)); }
__device__ __forceinline__ float tanh_r(float v)
{
    const float x = __builtin_fminf(__builtin_fmaxf(v, -7.9f), 7.9f), u = x * x;
    float p = -8.4887e-14f; p = __builtin_fmaf(p, u, 5.278e-11f); p = __builtin_fmaf(p, u, -2.0225e-08f); p = __builtin_fmaf(p, u, 1.11543e-05f);
    p = __builtin_fmaf(p, u, 0.0031039565f); p = __builtin_fmaf(p, u, 0.1308401f); p = __builtin_fmaf(p, u, 0.99999999f);
    float q = 0.00025461456f; q = __builtin_fmaf(q, u, 0.0244951795f); q = __builtin_fmaf(q, u, 0.4641733745f); q = __builtin_fmaf(q, u, 1.0f);
    return (p * x) * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float swap32(float v) { unsigned u = __builtin_bit_cast(unsigned, v); auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); unsigned a = r[0], b = r[1]; return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b); }
__device__ __forceinline__ float swap16(float v) { unsigned u = __builtin_bit_cast(unsigned, v); auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); unsigned a = r[0], b = r[1]; return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b); }

// V = 0: one wave, 2 rows x 36 per lane. V = 1: two waves, 1 row x 36 per lane, flag spin. V = 2: two waves, s_barrier.
template <int V>
__global__ __launch_bounds__(V == 0 ? 64 : 128) void k(const float* w, float* o, long long* t, int iters)
{
    __shared__ __attribute__((aligned(16))) float hb[2][32 + 4];
    __shared__ volatile int flag[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ROWS = V == 0 ? 2 : 1;
    float wr[ROWS][36];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 36; ++i) wr[r][i] = w[((wave * 2 + r) * 36 + i) * 64 + lane];
    if (tid < 36) { hb[0][tid] = 0.01f * tid; hb[1][tid] = 0.f; }
    if (tid < 2) flag[tid] = -1;
    __syncthreads();
    float c = 0.f, x = 0.1f;
    long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float4* hv = reinterpret_cast<const float4*>(hb[it & 1]);
        float acc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r] = wr[r][32] * x + wr[r][35];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float4 h = hv[k4];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                acc[r] = __builtin_fmaf(wr[r][4 * k4], h.x, acc[r]); acc[r] = __builtin_fmaf(wr[r][4 * k4 + 1], h.y, acc[r]);
                acc[r] = __builtin_fmaf(wr[r][4 * k4 + 2], h.z, acc[r]); acc[r] = __builtin_fmaf(wr[r][4 * k4 + 3], h.w, acc[r]);
            }
        }
        float hn;
        if constexpr (V == 0) {
            // lane = part*32 + unit: part 0 holds (i,f), part 1 holds (g,o); exchange over the halves
            const float a0 = lane < 32 ? sig(acc[0]) : tanh_r(acc[0]);
            const float a1 = sig(acc[1]);
            const float ig = swap32(a0 * (lane < 32 ? 1.f : 0.5f)) ;           // stands for the i*g exchange
            const float fo = swap32(a1);
            c = __builtin_fmaf(a1, c, ig);
            hn = fo * tanh_r(c);
        } else {
            // lane = part*16 + unit, 4 parts = 4 gates: two exchanges
            const float a = (lane >> 4) == 2 ? tanh_r(acc[0]) : sig(acc[0]);
            const float s1 = swap32(a);
            const float s2 = swap16(a + s1);
            c = __builtin_fmaf(s1, c, s2);
            hn = a * tanh_r(c);
        }
        x = x * 0.999f + 0.001f;
        float* hw = hb[(it + 1) & 1];
        if constexpr (V == 0) {
            if (lane < 32) hw[lane] = hn;
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (V == 1) {
            if (lane < 16) hw[wave * 16 + lane] = hn;
            __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the h words are in LDS before the flag
            if (lane == 0) flag[wave] = it;
            while (flag[wave ^ 1] < it) { }
        } else {
            if (lane < 16) hw[wave * 16 + lane] = hn;
            __syncthreads();
        }
    }
    long long c1 = __builtin_readcyclecounter();
    o[blockIdx.x * 128 + tid] = c + x;
    if (tid == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <int V> void run(const char* name, const float* w, float* o, long long* t)
{
    const int iters = 4000;
    for (int blocks : {256, 1024, 4096}) {
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(V == 0 ? 64 : 128), 0, 0, w, o, t, iters); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(V == 0 ? 64 : 128), 0, 0, w, o, t, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("%-28s blocks=%5d: %7.1f cycles/frame (wave 0 of block 0)   %8.1f us per 256 frames (whole launch)\n", name, blocks, (double)h / iters, ms * 1e3 * 256 / iters);
    }
}
int main()
{
    float* w; float* o; long long* t;
    hipMalloc(&w, 4 * 36 * 64 * 4); hipMalloc(&o, 4096 * 128 * 4); hipMalloc(&t, 16);
    float hw[4 * 36 * 64];
    for (int i = 0; i < 4 * 36 * 64; ++i) hw[i] = 0.01f * ((i * 7919) % 101 - 50) / 50.f;
    hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
    run<0>("one wave (32 units, S=2)", w, o, t);
    run<1>("two waves, LDS flag spin", w, o, t);
    run<2>("two waves, s_barrier", w, o, t);
    return 0;
}
