// How many plugin instances does one GPU carry in real time through aidax_hub? A sequential "host" calls
// run() of N instances per 256-frame period; we time whole periods (staging + launch + read-back).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/aidax.h"
int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "tests/golden/models/tw40_california_clean_deerinkstudios.json";
    aidax_model* m = nullptr;
    if (aidax_model_load(path, &m) != AIDAX_OK) { printf("%s\n", aidax_last_error()); return 1; }
    for (int N : {64, 1024, 4096, 16384}) {
        const uint32_t n = 256;
        aidax_hub* hub = nullptr;
        if (aidax_hub_create(N, n, 48000.0, 0, &hub) != AIDAX_OK) { printf("%s\n", aidax_last_error()); return 1; }
        aidax_hub_set_model(hub, m, AIDAX_START_WARMUP);
        std::vector<int32_t> slot(N);
        for (int i = 0; i < N; ++i) aidax_hub_attach(hub, &slot[i]);
        std::vector<float> in(n), out(n);
        for (uint32_t t = 0; t < n; ++t) in[t] = 0.1f * ((t * 37) % 17 - 8) / 8.f;
        const int periods = 60;
        for (int p = 0; p < 5; ++p) for (int i = 0; i < N; ++i) aidax_hub_run(hub, slot[i], in.data(), out.data(), n);
        const auto t0 = std::chrono::steady_clock::now();
        for (int p = 0; p < periods; ++p) for (int i = 0; i < N; ++i) aidax_hub_run(hub, slot[i], in.data(), out.data(), n);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / periods;
        printf("%6d instances: %9.1f us per 256-frame period (real time allows 5333 us) -> %5.1f %% of real time, launches %llu\n",
               N, us, 100.0 * us / 5333.3, (unsigned long long)aidax_hub_launches(hub));
        aidax_hub_destroy(hub);
    }
    aidax_model_free(m);
    return 0;
}
