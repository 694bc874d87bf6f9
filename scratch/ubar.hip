#include <hip/hip_runtime.h>
#include <cstdio>
// Cost of a per-iteration workgroup barrier + LDS exchange between W waves, with X dependent FMAs of work per iteration.
template <int W, int X>
__global__ __launch_bounds__(W * 64) void k(float* o, long long* t, int iters) {
  __shared__ float xb[W * 64];
  const int tid = threadIdx.x;
  float a = tid * 1e-3f;
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < X; ++j) a = __builtin_fmaf(a, 1.0001f, 1e-7f);
    xb[tid] = a;
    __syncthreads();
    a += xb[(tid + 64) % (W * 64)];
    __syncthreads();   // WAR protection (second barrier) -- variant B below uses double buffering instead
  }
  long long c1 = __builtin_readcyclecounter();
  o[blockIdx.x * W * 64 + tid] = a;
  if (tid == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <int W, int X>
__global__ __launch_bounds__(W * 64) void k2(float* o, long long* t, int iters) {
  __shared__ float xb[2][W * 64];
  const int tid = threadIdx.x;
  float a = tid * 1e-3f;
  long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < X; ++j) a = __builtin_fmaf(a, 1.0001f, 1e-7f);
    xb[it & 1][tid] = a;
    __syncthreads();
    a += xb[it & 1][(tid + 64) % (W * 64)];
  }
  long long c1 = __builtin_readcyclecounter();
  o[blockIdx.x * W * 64 + tid] = a;
  if (tid == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
template <int W, int X> void run(float* o, long long* t, int blocks) {
  const int iters = 2000;
  k<W, X><<<blocks, W * 64>>>(o, t, iters); hipDeviceSynchronize();
  k<W, X><<<blocks, W * 64>>>(o, t, iters); hipDeviceSynchronize();
  long long h1; hipMemcpy(&h1, t, 8, hipMemcpyDeviceToHost);
  k2<W, X><<<blocks, W * 64>>>(o, t, iters); hipDeviceSynchronize();
  k2<W, X><<<blocks, W * 64>>>(o, t, iters); hipDeviceSynchronize();
  long long h2; hipMemcpy(&h2, t, 8, hipMemcpyDeviceToHost);
  printf("W=%d X=%3d blocks=%5d: 2-barrier %7.1f cyc/iter   1-barrier(dbuf) %7.1f cyc/iter\n", W, X, blocks, (double)h1 / iters, (double)h2 / iters);
}
int main() {
  float* o; long long* t; hipMalloc(&o, 64 << 20); hipMalloc(&t, 16);
  for (int blocks : {256, 1024}) {
    run<2, 0>(o, t, blocks); run<2, 40>(o, t, blocks); run<2, 80>(o, t, blocks);
    run<4, 0>(o, t, blocks); run<4, 40>(o, t, blocks); run<4, 80>(o, t, blocks);
  }
}
