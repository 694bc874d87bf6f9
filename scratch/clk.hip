#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o, long long* t, int iters) {
  float a = threadIdx.x * 1e-9f, b = 1.000001f;
  long long c0 = __builtin_readcyclecounter(); long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) a = __builtin_fmaf(a, b, 1e-7f);
  }
  long long c1 = __builtin_readcyclecounter(); long long w1 = wall_clock64();
  o[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  float* o; long long* t; hipMalloc(&o, 4 << 20); hipMalloc(&t, 16);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("wallclock rate %d kHz, clockRate %d kHz\n", rate, clk);
  for (int blocks : {1, 256, 1024, 4096}) for (int iters : {2000, 20000}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 64>>>(o, t, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 64>>>(o, t, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    double nf = 64.0 * iters;
    printf("blocks %5d iters %6d: %8.3f ms  memtime %lld (%.2f ticks/fma) wall %lld  => shader clk %.0f MHz if 4cyc/fma; memtime rate %.1f MHz\n",
      blocks, iters, ms, h[0], h[0]/nf, h[1], nf*4/(ms*1e3), h[0]/(ms*1e3));
  }
}
