// Time of aidax_pool_submit and aidax_pool_collect, call by call (cfg2 block, pageable buffers), next to aidax_pool_process.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/aidax.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    aidax_model* m = nullptr;
    if (aidax_model_load(argv[1], &m) != AIDAX_OK) { printf("load: %s\n", aidax_last_error()); return 1; }
    aidax_pool* p = nullptr;
    const uint32_t S = 1024, n = 256;
    if (aidax_pool_create(S, n, 48000.0, 0, &p) != AIDAX_OK || aidax_pool_set_model(p, m, AIDAX_START_WARMUP) != AIDAX_OK) { printf("%s\n", aidax_last_error()); return 1; }
    std::vector<float> x(S * n), y(S * n);
    for (size_t i = 0; i < x.size(); ++i) x[i] = 0.3f * ((i * 7919u) % 1000) / 1000.f - 0.15f;
    for (int i = 0; i < 50; ++i) aidax_pool_process(p, x.data(), y.data(), n);
    const int N = 500;
    double t0 = now();
    for (int i = 0; i < N; ++i) aidax_pool_process(p, x.data(), y.data(), n);
    printf("aidax_pool_process: %.1f us per block\n", (now() - t0) / N * 1e6);
    aidax_pool_submit(p, x.data(), n);
    for (int i = 0; i < 50; ++i) { aidax_pool_submit(p, x.data(), n); aidax_pool_collect(p, y.data(), n); }
    double ts = 0, tc = 0;
    t0 = now();
    for (int i = 0; i < N; ++i) {
        const double a = now();
        aidax_pool_submit(p, x.data(), n);
        const double b = now();
        aidax_pool_collect(p, y.data(), n);
        ts += b - a; tc += now() - b;
    }
    printf("submit + collect: %.1f us per block (submit %.1f, collect %.1f)\n", (now() - t0) / N * 1e6, ts / N * 1e6, tc / N * 1e6);
    aidax_pool_collect(p, y.data(), n);
    // two blocks in flight: collect never waits (one more block of latency)
    aidax_pool_submit(p, x.data(), n); aidax_pool_submit(p, x.data(), n);
    for (int i = 0; i < 50; ++i) { aidax_pool_collect(p, y.data(), n); aidax_pool_submit(p, x.data(), n); }
    ts = tc = 0;
    t0 = now();
    for (int i = 0; i < N; ++i) {
        const double a = now();
        aidax_pool_collect(p, y.data(), n);
        const double b = now();
        aidax_pool_submit(p, x.data(), n);
        tc += b - a; ts += now() - b;
    }
    printf("two in flight:    %.1f us per block (submit %.1f, collect %.1f)\n", (now() - t0) / N * 1e6, ts / N * 1e6, tc / N * 1e6);
    aidax_pool_collect(p, y.data(), n); aidax_pool_collect(p, y.data(), n);
    // the caller's buffers page-locked once (aidax_pool_register_host): upload straight out of x, download straight into y0 / y1
    std::vector<float> y0(S * n), y1(S * n), yref(S * n);
    aidax_pool_process(p, x.data(), yref.data(), n);        // (the recurrent state moves on: compare like with like below)
    if (aidax_pool_register_host(p, x.data(), x.size() * 4) != AIDAX_OK || aidax_pool_register_host(p, y0.data(), y0.size() * 4) != AIDAX_OK ||
        aidax_pool_register_host(p, y1.data(), y1.size() * 4) != AIDAX_OK) { printf("register: %s\n", aidax_last_error()); return 1; }
    float* yy[2] = { y0.data(), y1.data() };
    aidax_pool_submit_to(p, x.data(), yy[0], n);
    for (int i = 1; i <= 50; ++i) { aidax_pool_submit_to(p, x.data(), yy[i & 1], n); aidax_pool_collect(p, yy[(i - 1) & 1], n); }
    ts = tc = 0;
    t0 = now();
    for (int i = 51; i < 51 + N; ++i) {
        const double a = now();
        aidax_pool_submit_to(p, x.data(), yy[i & 1], n);
        const double b = now();
        aidax_pool_collect(p, yy[(i - 1) & 1], n);
        ts += b - a; tc += now() - b;
    }
    printf("registered buffers, submit_to + collect: %.1f us per block (submit %.1f, collect %.1f)\n", (now() - t0) / N * 1e6, ts / N * 1e6, tc / N * 1e6);
    aidax_pool_collect(p, yy[(51 + N - 1) & 1], n);
    double mx = 0; for (size_t i = 0; i < y0.size(); ++i) { const double d = y0[i] > 0 ? y0[i] : -y0[i]; if (d > mx) mx = d; }
    printf("(last block: max |y| = %.4f, finite: %d)\n", mx, mx == mx && mx < 100.0);
    aidax_pool_unregister_host(p, x.data()); aidax_pool_unregister_host(p, y0.data()); aidax_pool_unregister_host(p, y1.data());
    aidax_pool_destroy(p);
    aidax_model_free(m);
    return 0;
}
