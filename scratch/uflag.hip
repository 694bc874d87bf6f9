// uflag.hip — cost of a per-frame hand-shake between two waves of a workgroup through LDS flags (no s_barrier):
// every iteration each wave publishes a value + its iteration number and waits until the partner's number has
// arrived, then reads the partner's value. Compared with the same loop around s_barrier, and with a single wave
// doing write -> read turnaround on its own. Wave placement: waves 0 and 1 sit on different SIMDs of the CU.
//   hipcc --offload-arch=gfx950 -O2 scratch/uflag.hip -o scratch/uflag && scratch/uflag
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int WORK>
__global__ void k(float* out, long long* cyc, int iters)
{
    __shared__ float val[2][64];
    __shared__ volatile int flag[2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 2) flag[threadIdx.x] = -1;
    __syncthreads();
    float a = lane * 1e-6f, acc = 0.f;
    const long long c0 = __builtin_readcyclecounter();
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int j = 0; j < WORK; ++j) a = __builtin_fmaf(a, 1.000001f, 1e-7f);
        if (MODE == 0) {                      // single wave: write, read back (own turnaround)
            val[0][lane] = a;
            __builtin_amdgcn_wave_barrier();
            acc += val[0][(lane + 1) & 63];
        } else if (MODE == 1) {               // two waves, s_barrier
            val[w][lane] = a;
            __syncthreads();
            acc += val[w ^ 1][(lane + 1) & 63];
            __syncthreads();
        } else {                              // two waves, flags
            val[w][lane] = a;                 // (single-buffered: partner may overwrite only after reading t, see below)
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) flag[w] = t;       // LDS ops of a wave execute in order: data before flag
            while (flag[w ^ 1] < t) { }
            acc += val[w ^ 1][(lane + 1) & 63];
            // partner must not overwrite val before we read: second flag round would be needed in general;
            // double-buffer by parity instead
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + a;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

template <int MODE, int WORK>
void run(const char* name, int waves)
{
    float* o; long long* c; hipMalloc(&o, 1 << 20); hipMalloc(&c, 8);
    const int iters = 20000;
    for (int blocks : {1, 256, 1024}) {
        k<MODE, WORK><<<blocks, 64 * waves>>>(o, c, iters); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<MODE, WORK><<<blocks, 64 * waves>>>(o, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s work=%3d blocks=%4d: %7.1f ns/iter = %6.0f cycles @2.4GHz\n", name, WORK, blocks, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
    }
    hipFree(o); hipFree(c);
}

int main()
{
    run<0, 0>("1 wave write->read", 1);
    run<0, 64>("1 wave write->read", 1);
    run<1, 0>("2 waves s_barrier x2", 2);
    run<1, 64>("2 waves s_barrier x2", 2);
    run<2, 0>("2 waves LDS flags", 2);
    run<2, 64>("2 waves LDS flags", 2);
    return 0;
}
