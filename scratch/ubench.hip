#include <hip/hip_runtime.h>
#include <cstdio>
// Per-iteration cost of a 2x32 recurrent dot product under different h-broadcast schemes (1 wave per block).
#define FM(a,b,c) __builtin_fmaf(a,b,c)
template <int K> __device__ __forceinline__ float rowb(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, true));
}
#define DPPF(acc, h, w, K) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w))
template <int V>
__global__ __launch_bounds__(64) void k(float* o, const float* w, long long* t, int iters) {
  __shared__ float hb[64];
  const int lane = threadIdx.x;
  float wa[32], wb[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { wa[i] = w[i * 64 + lane]; wb[i] = w[(32 + i) * 64 + lane]; }
  float h = lane * 1e-3f, h2 = h * 0.5f;
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float a = 0.1f, b = 0.2f;
    if constexpr (V == 0) {          // registers only (no broadcast): lower bound
#pragma unroll
      for (int i = 0; i < 32; ++i) { a = FM(wa[i], h, a); b = FM(wb[i], h2, b); }
    } else if constexpr (V == 1) {   // asm fmac_dpp row_newbcast
      asm volatile("s_nop 1");
#define STEP(i) DPPF(a, h, wa[i], i); DPPF(b, h, wb[i], i);
#define STEP2(i) DPPF(a, h2, wa[16+i], i); DPPF(b, h2, wb[16+i], i);
      STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
      STEP2(0) STEP2(1) STEP2(2) STEP2(3) STEP2(4) STEP2(5) STEP2(6) STEP2(7) STEP2(8) STEP2(9) STEP2(10) STEP2(11) STEP2(12) STEP2(13) STEP2(14) STEP2(15)
    } else if constexpr (V == 2) {   // compiler mov_dpp + fma
#define C(i) { float q = rowb<i>(h); a = FM(wa[i], q, a); b = FM(wb[i], q, b); float r = rowb<i>(h2); a = FM(wa[16+i], r, a); b = FM(wb[16+i], r, b); }
      C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
    } else if constexpr (V == 3) {   // LDS broadcast
      if (lane < 32) hb[lane] = h;
      __builtin_amdgcn_wave_barrier();
      const float4* hv = (const float4*)hb;
#pragma unroll
      for (int i = 0; i < 8; ++i) { float4 q = hv[i]; a = FM(wa[4*i], q.x, a); b = FM(wb[4*i], q.x, b); a = FM(wa[4*i+1], q.y, a); b = FM(wb[4*i+1], q.y, b);
        a = FM(wa[4*i+2], q.z, a); b = FM(wb[4*i+2], q.z, b); a = FM(wa[4*i+3], q.w, a); b = FM(wb[4*i+3], q.w, b); }
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (V == 4) {   // packed fma: (a,b) pair, weights pair, h broadcast by op_sel
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 acc = {a, b};
#pragma unroll
      for (int i = 0; i < 32; ++i) { f2 ww = {wa[i], wb[i]}; f2 hh = {h, h2};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(ww), "v"(hh)); }
      a = acc.x; b = acc.y;
    } else if constexpr (V == 6 || V == 7) {   // packed fma, V==6: two / V==7: four independent accumulator pairs
      typedef float f2 __attribute__((ext_vector_type(2)));
      constexpr int NC = V == 6 ? 2 : 4;
      f2 acc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = f2{a, b};
#pragma unroll
      for (int i = 0; i < 32; ++i) { f2 ww = {wa[i], wb[i]}; f2 hh = {h, h2};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i % NC]) : "v"(ww), "v"(hh)); }
      f2 sum = acc[0];
#pragma unroll
      for (int c = 1; c < NC; ++c) sum += acc[c];
      a = sum.x; b = sum.y;
    } else if constexpr (V == 8) {   // scalar fma, four independent chains (2 rows x even/odd k)
      float a2 = 0.f, b2 = 0.f;
#pragma unroll
      for (int i = 0; i < 32; i += 2) { a = FM(wa[i], h, a); b = FM(wb[i], h2, b); a2 = FM(wa[i + 1], h, a2); b2 = FM(wb[i + 1], h2, b2); }
      a += a2; b += b2;
    } else if constexpr (V == 5) {   // readlane -> SGPR broadcast
#pragma unroll
      for (int i = 0; i < 32; ++i) { float q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), i)); a = FM(wa[i], q, a); b = FM(wb[i], q, b); }
    }
    h = a * 0.01f + b * 0.001f; h2 = a - b;
  }
  long long c1 = __builtin_readcyclecounter();
  o[blockIdx.x * 64 + lane] = h + h2;
  if (lane == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <int V> void run(const char* name, float* o, float* w, long long* t) {
  for (int blocks : {1024, 4096}) {
    const int iters = 2000;
    k<V><<<blocks, 64>>>(o, w, t, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<V><<<blocks, 64>>>(o, w, t, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-28s blocks %5d: %7.1f cycles/iter (memtime)  %8.3f ms -> %6.1f ns/iter\n", name, blocks, (double)h / iters, ms, ms * 1e6 / iters);
  }
}
int main() {
  float *o, *w; long long* t; hipMalloc(&o, 4 << 20); hipMalloc(&w, 64 * 64 * 4); hipMalloc(&t, 16);
  hipMemset(w, 0, 64 * 64 * 4);
  run<0>("regs only (no bcast)", o, w, t);
  run<1>("asm fmac_dpp newbcast", o, w, t);
  run<2>("compiler mov_dpp + fma", o, w, t);
  run<3>("LDS write + b128 bcast", o, w, t);
  run<4>("v_pk_fma_f32", o, w, t);
  run<5>("readlane->sgpr", o, w, t);
  run<6>("v_pk_fma_f32 x2 chains", o, w, t);
  run<7>("v_pk_fma_f32 x4 chains", o, w, t);
  run<8>("v_fma_f32 x4 chains", o, w, t);
}
