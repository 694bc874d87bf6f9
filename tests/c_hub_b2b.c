/* c_hub_b2b.c — drives aidax_hub from C with no gaps between the calls (tests/test_hub.py): attach, activate,
 * controls and run() of a freshly attached instance follow each other back to back, while the pass of the period
 * before is still in flight on the hub's stream. The pool's control pokes (reset_stream, warm-up, activate) run
 * on the pool's own stream and the passes on the hub's: the result is only right if the pool orders the two
 * (stream edges in aidax_pool.cpp). Writes every instance's output rows to a raw float file for the test to
 * compare with the oracle.
 *
 *   c_hub_b2b model.json out.f32 periods frames
 * Instance A attaches at period 0, B at period 2, C at period 3 (B and C join while passes are in flight). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aidax.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != AIDAX_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, aidax_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    aidax_model* m = NULL;
    aidax_hub* hub = NULL;
    aidax_controls c;
    int32_t slot[3] = { -1, -1, -1 };
    const int join_at[3] = { 0, 2, 3 };
    int periods, n, p, i, t;
    float *x, *y;
    FILE* f;
    if (argc < 5) return 2;
    periods = atoi(argv[3]);
    n = atoi(argv[4]);
    CHECK(aidax_model_load(argv[1], &m));
    CHECK(aidax_hub_create(8, (uint32_t)n, 48000.0, 0, &hub));
    CHECK(aidax_hub_set_deadline_us(hub, 0));
    CHECK(aidax_hub_set_model(hub, m, AIDAX_START_WARMUP));
    x = (float*)malloc(sizeof(float) * (size_t)n);
    y = (float*)calloc((size_t)3 * periods * n, sizeof(float));
    if (!x || !y) return 3;
    aidax_controls_default(&c);
    for (p = 0; p < periods; ++p) {
        for (i = 0; i < 3; ++i) {
            if (p < join_at[i]) continue;
            if (p == join_at[i]) {
                CHECK(aidax_hub_attach(hub, &slot[i]));
                c.pregain_db = (float)(2 * i);            /* each instance its own controls */
                c.param1 = 0.25f * (float)i;
                CHECK(aidax_hub_set_controls(hub, slot[i], &c));
            }
            /* deterministic input the test regenerates: a per-instance pseudo-random sequence */
            for (t = 0; t < n; ++t) {
                const unsigned k = (unsigned)(p * n + t) * 2654435761u + (unsigned)(i + 1) * 40503u;
                x[t] = ((float)((k >> 8) & 0xffff) / 65535.0f - 0.5f) * 0.8f;
            }
            CHECK(aidax_hub_run(hub, slot[i], x, y + ((size_t)i * periods + p) * n, (uint32_t)n));
        }
    }
    f = fopen(argv[2], "wb");
    if (!f) return 4;
    fwrite(y, sizeof(float), (size_t)3 * periods * n, f);
    fclose(f);
    printf("launches=%llu\n", (unsigned long long)aidax_hub_launches(hub));
    aidax_hub_destroy(hub);
    aidax_model_free(m);
    free(x); free(y);
    return 0;
}
