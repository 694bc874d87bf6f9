// What does one fp64 VALU instruction cost a LONE wave (one wave per SIMD: the chain passes' regime)? Dependent chains against
// independent ones, fma / mul / add, and the biquad step itself (Biquad.h:53-58, nine operations, no contraction) as the
// compiler schedules it. Wall time per iteration in ns and in shader cycles (s_memrealtime, 100 MHz, times the measured clock).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize scratch/ufp64.hip -o /tmp/ufp64 && /tmp/ufp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Rec { unsigned long long c0, c1, w0, w1; };
template <int MODE>
__global__ void k(Rec* out, double* sink, int iters, double b, double c)
{
    double v0 = threadIdx.x * 1e-3, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    double z1 = 0, z2 = 0, y1 = 0, y2 = 0;
    const double a0 = b, a1 = c, a2 = b * 0.5, b1 = c * 0.25, b2 = b * 0.125;
    // per-lane coefficients (VGPRs, like the chain's): nothing the compiler can keep in SGPRs
    const double va0 = b + threadIdx.x * 1e-9, va1 = c + threadIdx.x * 1e-9, va2 = va0 * 0.5, vb1 = va1 * 0.25, vb2 = va0 * 0.125;
    const float gf = 1.f + threadIdx.x * 1e-7f;
    const float fb_s = (float)b, fb_v = (float)b + threadIdx.x * 1e-9f, fc_v = (float)c + threadIdx.x * 1e-9f;
    float fv[8];
    for (int j = 0; j < 8; ++j) fv[j] = threadIdx.x * 1e-3f + j;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {                 // 8 dependent fma
#pragma unroll
            for (int j = 0; j < 8; ++j) v0 = __builtin_fma(v0, b, c);
        } else if (MODE == 1) {          // 8 independent fma
            v0 = __builtin_fma(v0, b, c); v1 = __builtin_fma(v1, b, c); v2 = __builtin_fma(v2, b, c); v3 = __builtin_fma(v3, b, c);
            v4 = __builtin_fma(v4, b, c); v5 = __builtin_fma(v5, b, c); v6 = __builtin_fma(v6, b, c); v7 = __builtin_fma(v7, b, c);
        } else if (MODE == 2) {          // 8 dependent mul/add alternating
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0 = v0 * b; v0 = v0 + c; }
        } else if (MODE == 3) {          // 8 independent: 4 mul + 4 add
            v0 = v0 * b; v1 = v1 + c; v2 = v2 * b; v3 = v3 + c; v4 = v4 * b; v5 = v5 + c; v6 = v6 * b; v7 = v7 + c;
        } else if (MODE == 4) {          // one biquad step, input from a cheap recurrence
            const double in = v0; v0 = v0 + c;
            const double o = in * a0 + z1;
            z1 = in * a1 + z2 - b1 * o;
            z2 = in * a2 - b2 * o;
            v1 += o;
        } else if (MODE == 5) {          // two independent biquads in one instruction stream
            const double in = v0; v0 = v0 + c;
            const double o = in * a0 + z1;
            z1 = in * a1 + z2 - b1 * o;
            z2 = in * a2 - b2 * o;
            const double p = in * a1 + y1;
            y1 = in * a0 + y2 - b2 * p;
            y2 = in * a2 - b1 * p;
            v1 += o + p;
        } else if (MODE == 7) {          // the chain's PLAIN macro-step as written (aidax_device.h): 8 samples per block, 12 ops each
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (float)v0 + (float)q;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double xd = v[q];
                const double yd = xd * va0 + z1;
                z1 = xd * va1 + z2 - vb1 * yd;
                z2 = xd * va2 - vb2 * yd;
                v[q] = (float)yd * gf;
            }
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
            v0 = acc;
        } else if (MODE == 8) {          // the same values, the input-only products of the whole block first
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (float)v0 + (float)q;
            double m0[8], m1[8], m2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double xd = v[q]; m0[q] = xd * va0; m1[q] = xd * va1; m2[q] = xd * va2; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double yd = m0[q] + z1;
                z1 = m1[q] + z2 - vb1 * yd;
                z2 = m2[q] - vb2 * yd;
                v[q] = (float)yd * gf;
            }
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
            v0 = acc;
        } else if (MODE == 9) {          // fp32: 8 dependent fma, operands in VGPRs
            float f = (float)v0;
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(fb_v), "v"(fc_v));
            v0 = f;
        } else if (MODE == 10) {         // fp32: 8 dependent fma, one operand an SGPR
            float f = (float)v0;
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "s"(fb_s), "v"(fc_v));
            v0 = f;
        } else if (MODE == 11) {         // fp32: 8 independent fma, VGPRs
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[j]) : "v"(fb_v), "v"(fc_v));
        } else if (MODE == 12) {         // fp32: 8 independent fma, one operand an SGPR
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fv[j]) : "s"(fb_s), "v"(fc_v));
        } else if (MODE == 13) {         // fp32: 8 independent fma, an inline constant operand
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(fv[j]) : "v"(fc_v));
        } else if (MODE == 14) {         // fp64: 8 independent mul, VGPR operands
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v0) : "v"(va0)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v1) : "v"(va0));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v2) : "v"(va0)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v3) : "v"(va0));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v4) : "v"(va0)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v5) : "v"(va0));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v6) : "v"(va0)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v7) : "v"(va0));
        } else if (MODE == 15) {         // fp64: 8 independent mul, one SGPR operand
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v0) : "s"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v1) : "s"(b));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v2) : "s"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v3) : "s"(b));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v4) : "s"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v5) : "s"(b));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v6) : "s"(b)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v7) : "s"(b));
        } else if (MODE == 16) {         // fp64: 8 independent fma, VGPR operands
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(va0), "v"(va1)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v1) : "v"(va0), "v"(va1));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v2) : "v"(va0), "v"(va1)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v3) : "v"(va0), "v"(va1));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v4) : "v"(va0), "v"(va1)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v5) : "v"(va0), "v"(va1));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v6) : "v"(va0), "v"(va1)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v7) : "v"(va0), "v"(va1));
        } else if (MODE == 6) {          // 8 dependent fp32 fma (reference point)
            float f = (float)v0;
#pragma unroll
            for (int j = 0; j < 8; ++j) f = __builtin_fmaf(f, (float)b, (float)c);
            v0 = f;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{c0, c1, w0, w1};
    float fs = 0.f;
    for (int j = 0; j < 8; ++j) fs += fv[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + z1 + z2 + y1 + y2 + fs;
}
template <int MODE>
void run(const char* name, int blocks, int threads, int iters, int ops)
{
    Rec* d; double* s;
    hipMalloc(&d, blocks * sizeof(Rec)); hipMalloc(&s, (size_t)blocks * threads * sizeof(double));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, s, iters, 0.999, 1e-3);
    hipDeviceSynchronize();
    std::vector<Rec> h(blocks);
    hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    double cyc = 0, ghz = 0;
    std::vector<double> per;
    for (auto& r : h) { cyc += (double)(r.c1 - r.c0); ghz += (double)(r.c1 - r.c0) / ((double)(r.w1 - r.w0) * 10.0); per.push_back((double)(r.c1 - r.c0) / iters); }
    cyc /= blocks; ghz /= blocks;
    std::sort(per.begin(), per.end());
    printf("%-52s %4d wgs x %3d thr: %7.1f cycles/iter = %5.2f per op  (clock %.2f GHz)  per-wg p10 %.1f median %.1f p90 %.1f\n", name, blocks, threads,
           cyc / iters, cyc / iters / ops, ghz, per[per.size() / 10], per[per.size() / 2], per[per.size() * 9 / 10]);
    hipFree(d); hipFree(s);
}
int main()
{
    for (int thr : { 64, 256, 512 }) {         // 1 wave per CU, 1 per SIMD, 2 per SIMD
        run<0>("8 dependent v_fma_f64", 256, thr, 20000, 8);
        run<1>("8 independent v_fma_f64", 256, thr, 20000, 8);
        run<2>("8 dependent v_mul_f64 / v_add_f64", 256, thr, 20000, 8);
        run<3>("8 independent v_mul_f64 / v_add_f64", 256, thr, 20000, 8);
        run<4>("one biquad step (9 ops + 2)", 256, thr, 20000, 11);
        run<5>("two independent biquad steps (18 ops + 3)", 256, thr, 20000, 21);
        run<6>("8 dependent v_fma_f32 (+2 cvt)", 256, thr, 20000, 10);
        run<7>("chain macro-step as written: per SAMPLE (12 ops)", 256, thr, 20000, 8);
        run<8>("... input products first: per SAMPLE", 256, thr, 20000, 8);
    }
    run<9>("fp32: 8 dependent v_fma_f32, VGPR operands (+2 cvt)", 256, 256, 20000, 8);
    run<10>("fp32: 8 dependent v_fma_f32, one SGPR operand (+2 cvt)", 256, 256, 20000, 8);
    run<11>("fp32: 8 independent v_fma_f32, VGPR operands", 256, 256, 20000, 8);
    run<12>("fp32: 8 independent v_fma_f32, one SGPR operand", 256, 256, 20000, 8);
    run<13>("fp32: 8 independent v_fma_f32, an inline constant", 256, 256, 20000, 8);
    run<14>("fp64: 8 independent v_mul_f64, VGPR operands", 256, 256, 20000, 8);
    run<15>("fp64: 8 independent v_mul_f64, one SGPR operand", 256, 256, 20000, 8);
    run<16>("fp64: 8 independent v_fma_f64, VGPR operands", 256, 256, 20000, 8);
    // the fused conv kernel's regime: four workgroups per CU, one busy wave each
    run<7>("chain macro-step, 4 one-wave workgroups per CU", 1024, 64, 20000, 8);
    run<8>("... input products first", 1024, 64, 20000, 8);
    run<7>("chain macro-step, 8 one-wave workgroups per CU", 2048, 64, 20000, 8);
    return 0;
}
