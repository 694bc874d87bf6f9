/* The public header is plain C (C99): this file is compiled with gcc -std=c99 -pedantic by
 * tests/test_abi_host.py and run without a GPU: model loading, info, goldens, control defaults, biquad
 * design work on the host; creating a pool reports AIDAX_ERR_DEVICE instead of falling back. */
#include <stdio.h>
#include <string.h>
#include "aidax.h"

int main(int argc, char** argv)
{
    aidax_model* m = NULL;
    aidax_model_info_t info;
    aidax_controls c;
    double bq[5];
    aidax_pool* pool = NULL;
    int rc;
    if (argc < 2) return 2;
    rc = aidax_model_load(argv[1], &m);
    if (rc != AIDAX_OK) { printf("load: %d %s\n", rc, aidax_last_error()); return 1; }
    memset(&info, 0, sizeof info);
    aidax_model_info(m, &info);
    printf("cell=%d hidden=%d inputs=%d layers=%d\n", info.cell, info.hidden, info.input_size, info.n_rnn_layers);
    aidax_controls_default(&c);
    printf("lpf=%.3f master=%.1f enabled=%.0f\n", c.in_lpf_pc, c.master_db, c.enabled);
    if (aidax_biquad_design(0, 0.25, 0.707, 0.0, bq) != AIDAX_OK) return 1;
    printf("lowpass a0=%.6f\n", bq[0]);
    {   /* the round-3 entry points from plain C: null handles are argument errors, never crashes */
        aidax_stream_dsp dsp;
        float blk[4] = { 0.f, 0.f, 0.f, 0.f };
        int32_t slot = -1;
        memset(&dsp, 0, sizeof dsp);
        printf("null handles: submit=%d collect=%d export=%d import=%d successor=%d adopt=%d hub_frames=%u\n",
               aidax_pool_submit(NULL, blk, 4), aidax_pool_collect(NULL, blk, 4), aidax_pool_export_stream_dsp(NULL, 0, &dsp),
               aidax_pool_import_stream_dsp(NULL, 0, &dsp), aidax_hub_attach_successor(NULL, NULL, -1, &slot),
               aidax_hub_adopt(NULL, 0, NULL, 0), aidax_hub_max_frames(NULL));
        printf("sizeof(aidax_stream_dsp)=%u\n", (unsigned)sizeof dsp);
        printf("null handles: register=%d unregister=%d submit_to=%d\n", aidax_pool_register_host(NULL, blk, sizeof blk),
               aidax_pool_unregister_host(NULL, blk), aidax_pool_submit_to(NULL, blk, blk, 4));
    }
    rc = aidax_pool_create(1, 256, 48000.0, 0, &pool);
    printf("pool_create rc=%d pool=%s\n", rc, pool ? "set" : "null");
    if (pool) aidax_pool_destroy(pool);
    aidax_model_free(m);
    return 0;
}
