// Where does a host-buffer block's time go? memcpy between pageable and pinned host memory of the three flavours,
// and hipMemcpyAsync of 1 MiB each way (cfg2 block: 1024 streams x 256 frames).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t B = 1 << 20;
    float* pg_a = (float*)malloc(B); float* pg_b = (float*)malloc(B);
    memset(pg_a, 1, B); memset(pg_b, 2, B);
    float* d; hipMalloc(&d, B);
    hipStream_t q; hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    const struct { const char* name; unsigned flags; } kinds[] = { { "hipHostMallocDefault", hipHostMallocDefault }, { "hipHostMallocNonCoherent", hipHostMallocNonCoherent },
                                                                   { "hipHostMallocCoherent", hipHostMallocCoherent }, { "hipHostMallocWriteCombined", hipHostMallocWriteCombined } };
    for (auto& k : kinds) {
        float* h = nullptr;
        if (hipHostMalloc((void**)&h, B, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); continue; }
        memset(h, 0, B);
        const int N = 200;
        double t0 = now();
        for (int i = 0; i < N; ++i) { pg_a[i] = (float)i; memcpy(h, pg_a, B); }
        double t_in = (now() - t0) / N;
        t0 = now();
        for (int i = 0; i < N; ++i) { hipMemcpyAsync(d, h, B, hipMemcpyHostToDevice, q); hipStreamSynchronize(q); }
        double t_up = (now() - t0) / N;
        t0 = now();
        for (int i = 0; i < N; ++i) { hipMemcpyAsync(h, d, B, hipMemcpyDeviceToHost, q); hipStreamSynchronize(q); }
        double t_dn = (now() - t0) / N;
        t0 = now();
        for (int i = 0; i < N; ++i) { hipMemcpyAsync(h, d, B, hipMemcpyDeviceToHost, q); hipStreamSynchronize(q); memcpy(pg_b, h, B); }
        double t_out = (now() - t0) / N - t_dn;
        printf("%-28s memcpy pageable->pinned %6.1f us | H2D+sync %6.1f us | D2H+sync %6.1f us | memcpy pinned->pageable (fresh from the GPU) %6.1f us\n",
               k.name, t_in * 1e6, t_up * 1e6, t_dn * 1e6, t_out * 1e6);
        hipHostFree(h);
    }
    double t0 = now();
    for (int i = 0; i < 200; ++i) { pg_a[i] = (float)i; memcpy(pg_b, pg_a, B); }
    printf("pageable -> pageable memcpy %6.1f us\n", (now() - t0) / 200 * 1e6);
    t0 = now();
    for (int i = 0; i < 200; ++i) { hipMemcpyAsync(d, pg_a, B, hipMemcpyHostToDevice, q); hipStreamSynchronize(q); }
    printf("hipMemcpyAsync from PAGEABLE H2D+sync %6.1f us\n", (now() - t0) / 200 * 1e6);
    t0 = now();
    for (int i = 0; i < 200; ++i) { hipMemcpyAsync(pg_b, d, B, hipMemcpyDeviceToHost, q); hipStreamSynchronize(q); }
    printf("hipMemcpyAsync to PAGEABLE D2H+sync %6.1f us\n", (now() - t0) / 200 * 1e6);
    return 0;
}
