#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned v = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  o[threadIdx.x] = r[0]; o[64+threadIdx.x] = r[1]; o[128+threadIdx.x] = q[0]; o[192+threadIdx.x] = q[1];
  unsigned a = v + 100, b = v + 200;
  auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[256+threadIdx.x] = r2[0]; o[320+threadIdx.x] = r2[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 384*4); k<<<1,64>>>(d); unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"p32 r0","p32 r1","p16 r0","p16 r1","p32(a,b) r0","p32(a,b) r1"};
  for (int j = 0; j < 6; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; i += 4) printf(" %u", h[j*64+i]); printf("\n"); }
}
