// lone-wave issue interval: v_fma_f32 against v_pk_fma_f32 (dependent on 8 independent accumulators, the matvec's shape), one wave per workgroup, one workgroup per CU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, const float* in, int iters)
{
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i]; p[i] = f2{ a[i], a[i] * 0.5f }; }
    const float w0 = in[512 + threadIdx.x], w1 = in[576 + threadIdx.x];
    const f2 w = { w0, w1 }, h = { w1, w0 };
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w0), "v"(w1));
                else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(w), "v"(h));
                else if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[0]) : "v"(w), "v"(h));          // ONE dependent chain
                else if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 1]) : "v"(w), "v"(h));      // two chains, alternating
                else if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 1]) : "v"(w0), "v"(w1));       // two plain chains, alternating (today's code)
                else if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(w0), "v"(w1));           // one plain dependent chain
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i & 3]) : "v"(w), "v"(h));   // four chains
            }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[(blockIdx.x & 1023) * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + 64 * 1024)[0] = t1 - t0;
}
int main()
{
    float *d_in, *d_out; hipMalloc(&d_in, 4096); hipMalloc(&d_out, 64 * 1024 * 4 + 64);
    hipMemset(d_in, 0, 4096);
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
        for (int mode = 0; mode < 7; ++mode) {
            const int blocks = 256 * 4 * waves_per_simd;        // one-wave workgroups: 4 (8) per CU
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                default: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters); break;
                }
                hipDeviceSynchronize();
            }
            long long c; hipMemcpy(&c, d_out + 64 * 1024, 8, hipMemcpyDeviceToHost);
            const char* names[7] = { "v_fma_f32, 8 chains", "v_pk_fma_f32, 8 chains", "v_pk_fma_f32, 1 chain", "v_pk_fma_f32, 2 chains", "v_fma_f32, 2 chains", "v_fma_f32, 1 chain", "v_pk_fma_f32 op_sel, 4 chains" };
            printf("%s, %d wave(s) per SIMD: %.2f clock64 ticks per instruction (clock64 = s_memtime)\n", names[mode], waves_per_simd, (double)c / (iters * 32.0));
        }
    return 0;
}
