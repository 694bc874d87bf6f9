/*
 * aidax_oracle_dsp.c — CPU oracle, DSP half (biquads, smoothers, 7-stage chain).
 *
 * TEST INFRASTRUCTURE ONLY (see aidax_oracle.h). Build with -ffp-contract=off:
 * every arithmetic step below is one IEEE-754 operation in the same order the
 * reference source spells it, so this file is bit-identical to the reference's
 * Biquad.cpp / ValueSmoother.hpp built without FMA contraction (checked against
 * oracle/_ref by tests/test_oracle.py::test_live_reference_library_when_present).
 */
#include "aidax_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ======================================================================
 * Biquad — transposed direct form II, double coefficients and state.
 * Follows common/Biquad.cpp:67-165 (design) and common/Biquad.h:53-58 (tick).
 * ==================================================================== */

static void biquad_design(orc_biquad* f)
{
    const double V = pow(10, fabs(f->peakGain) / 20.0);   /* Biquad.cpp:69 */
    const double K = tan(M_PI * f->Fc);                   /* Biquad.cpp:70 */
    const double Q = f->Q;
    const double KK = K * K;
    const double r2 = sqrt(2);
    double norm;

    switch (f->type) {
    case ORC_BQ_LOWPASS:                                   /* :72-79 */
        norm = 1 / (1 + K / Q + KK);
        f->a0 = KK * norm;
        f->a1 = 2 * f->a0;
        f->a2 = f->a0;
        f->b1 = 2 * (KK - 1) * norm;
        f->b2 = (1 - K / Q + KK) * norm;
        break;
    case ORC_BQ_HIGHPASS:                                  /* :81-88 */
        norm = 1 / (1 + K / Q + KK);
        f->a0 = 1 * norm;
        f->a1 = -2 * f->a0;
        f->a2 = f->a0;
        f->b1 = 2 * (KK - 1) * norm;
        f->b2 = (1 - K / Q + KK) * norm;
        break;
    case ORC_BQ_BANDPASS:                                  /* :90-97 */
        norm = 1 / (1 + K / Q + KK);
        f->a0 = K / Q * norm;
        f->a1 = 0;
        f->a2 = -f->a0;
        f->b1 = 2 * (KK - 1) * norm;
        f->b2 = (1 - K / Q + KK) * norm;
        break;
    case ORC_BQ_NOTCH:                                     /* :99-106 */
        norm = 1 / (1 + K / Q + KK);
        f->a0 = (1 + KK) * norm;
        f->a1 = 2 * (KK - 1) * norm;
        f->a2 = f->a0;
        f->b1 = f->a1;
        f->b2 = (1 - K / Q + KK) * norm;
        break;
    case ORC_BQ_PEAK: {                                    /* :108-125 */
        const double kq  = 1 / Q * K;       /* plain-Q term  */
        const double vkq = V / Q * K;       /* gained-Q term */
        if (f->peakGain >= 0) {             /* boost: numerator carries V */
            norm = 1 / (1 + kq + KK);
            f->a0 = (1 + vkq + KK) * norm;
            f->a1 = 2 * (KK - 1) * norm;
            f->a2 = (1 - vkq + KK) * norm;
            f->b1 = f->a1;
            f->b2 = (1 - kq + KK) * norm;
        } else {                            /* cut: denominator carries V */
            norm = 1 / (1 + vkq + KK);
            f->a0 = (1 + kq + KK) * norm;
            f->a1 = 2 * (KK - 1) * norm;
            f->a2 = (1 - kq + KK) * norm;
            f->b1 = f->a1;
            f->b2 = (1 - vkq + KK) * norm;
        }
        break;
    }
    case ORC_BQ_LOWSHELF: {                                /* :126-143 */
        const double s2v = sqrt(2 * V);
        const double vkk = V * K * K;
        if (f->peakGain >= 0) {
            norm = 1 / (1 + r2 * K + KK);
            f->a0 = (1 + s2v * K + vkk) * norm;
            f->a1 = 2 * (vkk - 1) * norm;
            f->a2 = (1 - s2v * K + vkk) * norm;
            f->b1 = 2 * (KK - 1) * norm;
            f->b2 = (1 - r2 * K + KK) * norm;
        } else {
            norm = 1 / (1 + s2v * K + vkk);
            f->a0 = (1 + r2 * K + KK) * norm;
            f->a1 = 2 * (KK - 1) * norm;
            f->a2 = (1 - r2 * K + KK) * norm;
            f->b1 = 2 * (vkk - 1) * norm;
            f->b2 = (1 - s2v * K + vkk) * norm;
        }
        break;
    }
    case ORC_BQ_HIGHSHELF: {                               /* :144-161 */
        const double s2v = sqrt(2 * V);
        if (f->peakGain >= 0) {
            norm = 1 / (1 + r2 * K + KK);
            f->a0 = (V + s2v * K + KK) * norm;
            f->a1 = 2 * (KK - V) * norm;
            f->a2 = (V - s2v * K + KK) * norm;
            f->b1 = 2 * (KK - 1) * norm;
            f->b2 = (1 - r2 * K + KK) * norm;
        } else {
            norm = 1 / (V + s2v * K + KK);
            f->a0 = (1 + r2 * K + KK) * norm;
            f->a1 = 2 * (KK - 1) * norm;
            f->a2 = (1 - r2 * K + KK) * norm;
            f->b1 = 2 * (KK - V) * norm;
            f->b2 = (V - s2v * K + KK) * norm;
        }
        break;
    }
    default:
        break;
    }
}

void orc_biquad_set(orc_biquad* f, int type, double Fc, double Q, double gain_db)
{
    /* setBiquad -> setPeakGain -> calcBiquad (Biquad.cpp:55-65); state untouched */
    f->type = type;
    f->Q = Q;
    f->Fc = Fc;
    f->peakGain = gain_db;
    biquad_design(f);
}

void orc_biquad_init(orc_biquad* f, int type, double Fc, double Q, double gain_db)
{
    /* 4-arg constructor, Biquad.cpp:32-35: design, then clear state */
    memset(f, 0, sizeof(*f));
    orc_biquad_set(f, type, Fc, Q, gain_db);
    f->z1 = f->z2 = 0.0;
}

float orc_biquad_process(orc_biquad* f, float in)
{
    /* Biquad.h:53-58 — float in, double math, float out */
    const double x = in;
    const double y = x * f->a0 + f->z1;
    f->z1 = x * f->a1 + f->z2 - f->b1 * y;
    f->z2 = x * f->a2 - f->b2 * y;
    return (float)y;
}

void orc_biquad_block(orc_biquad* f, float* out, const float* in, uint32_t n)
{
    /* applyBiquadFilter, rt-neural-generic.cpp:42-46 */
    for (uint32_t i = 0; i < n; ++i) out[i] = orc_biquad_process(f, in[i]);
}

/* ======================================================================
 * Smoothers — common/ValueSmoother.hpp
 * ==================================================================== */

static int f_differs(float a, float b) { return fabsf(a - b) >= FLT_EPSILON; }  /* d_isNotEqual :41-46 */

static void expsm_update(orc_expsm* s) { s->coef = expf(-1.f / (s->tau * s->sampleRate)); }  /* :147-151 */

void orc_expsm_init(orc_expsm* s) { memset(s, 0, sizeof(*s)); }                  /* :89-95 */

void orc_expsm_set_sample_rate(orc_expsm* s, float sr)                           /* :97-104 */
{
    if (f_differs(s->sampleRate, sr)) { s->sampleRate = sr; expsm_update(s); }
}

void orc_expsm_set_time_constant(orc_expsm* s, float t60)                        /* :106-115 */
{
    const float newTau = t60 * (float)(1.0 / 6.91);
    if (f_differs(s->tau, newTau)) { s->tau = newTau; expsm_update(s); }
}

void orc_expsm_set_target(orc_expsm* s, float t) { s->target = t; }              /* :127-130 */
void orc_expsm_clear_to_target(orc_expsm* s) { s->mem = s->target; }             /* :132-135 */

float orc_expsm_next(orc_expsm* s)                                               /* :142-145 */
{
    s->mem = s->mem * s->coef + s->target * (1.f - s->coef);
    return s->mem;
}

static void linsm_update(orc_linsm* s) { s->step = (s->target - s->mem) / (s->tau * s->sampleRate); } /* :236-240 */

void orc_linsm_init(orc_linsm* s) { memset(s, 0, sizeof(*s)); }                  /* :173-179 */

void orc_linsm_set_sample_rate(orc_linsm* s, float sr)                           /* :181-188 */
{
    if (f_differs(s->sampleRate, sr)) { s->sampleRate = sr; linsm_update(s); }
}

void orc_linsm_set_time_constant(orc_linsm* s, float tau)                        /* :190-197 */
{
    if (f_differs(s->tau, tau)) { s->tau = tau; linsm_update(s); }
}

void orc_linsm_set_target(orc_linsm* s, float t)                                 /* :209-216 */
{
    if (f_differs(s->target, t)) { s->target = t; linsm_update(s); }
}

void orc_linsm_clear_to_target(orc_linsm* s) { s->mem = s->target; }             /* :218-221 */

float orc_linsm_next(orc_linsm* s)                                               /* :229-234 */
{
    const float y0 = s->mem;
    const float dy = s->target - y0;
    s->mem = y0 + copysignf(fminf(fabsf(dy), fabsf(s->step)), dy);
    return s->mem;
}

float orc_db_co(float db) { return db > -90.0f ? powf(10.0f, db * 0.05f) : 0.0f; }

float orc_lpf_fc(float pc)
{
    /* MAP(x, 0, 100, 0.99f*0.5f, 0.25f*0.5f) in float, rt-neural-generic.h:167,178-179 */
    const float out_min = 0.99f * 0.5f, out_max = 0.25f * 0.5f;
    return ((pc - 0.0f) * (out_max - out_min) / (100.0f - 0.0f)) + out_min;
}

void* orc_calloc_lines(size_t bytes);   /* aidax_oracle_nn.c: cache-line padded calloc */

/* ======================================================================
 * DynamicModel mirror
 * ==================================================================== */

void orc_apply_model(orc_dynmodel* m, float* out, uint32_t n)
{
    /* rt-neural-generic.cpp:148-240; the six branches collapse to this loop */
    float x[3] = { 0.f, 0.f, 0.f };
    for (uint32_t i = 0; i < n; ++i) {
        out[i] *= m->input_gain;
        x[0] = out[i];
        if (m->input_size >= 2) x[1] = orc_linsm_next(&m->param1Coeff);
        if (m->input_size >= 3) x[2] = orc_linsm_next(&m->param2Coeff);
        const float y = orc_net_forward(m->net, x);
        if (m->input_skip) out[i] += y; else out[i] = y;
        out[i] *= m->output_gain;
    }
}

orc_dynmodel* orc_dynmodel_create(orc_net* net, int input_size, int input_skip,
                                  float input_gain, float output_gain, float samplerate,
                                  float old_param1, float old_param2, int warmup)
{
    orc_dynmodel* m = (orc_dynmodel*)orc_calloc_lines(sizeof(*m));
    m->net = net;
    orc_net_reset(net);                                    /* :1035 */
    m->input_size = input_size;
    m->input_skip = input_skip != 0;                       /* :1048 */
    m->input_gain = input_gain;
    m->output_gain = output_gain;
    m->samplerate = samplerate;
    orc_linsm_init(&m->param1Coeff);                       /* :1053-1056 */
    orc_linsm_set_sample_rate(&m->param1Coeff, samplerate);
    orc_linsm_set_time_constant(&m->param1Coeff, 0.1f);
    orc_linsm_set_target(&m->param1Coeff, old_param1);
    orc_linsm_clear_to_target(&m->param1Coeff);
    orc_linsm_init(&m->param2Coeff);                       /* :1057-1060 */
    orc_linsm_set_sample_rate(&m->param2Coeff, samplerate);
    orc_linsm_set_time_constant(&m->param2Coeff, 0.1f);
    orc_linsm_set_target(&m->param2Coeff, old_param2);
    orc_linsm_clear_to_target(&m->param2Coeff);
    m->paramFirstRun = 1;                                  /* :1061 */
    if (warmup) {                                          /* :1077-1078 */
        float zeros[2048];
        memset(zeros, 0, sizeof(zeros));
        orc_apply_model(m, zeros, 2048);
    }
    return m;
}

void orc_dynmodel_free(orc_dynmodel* m)
{
    if (!m) return;
    orc_net_free(m->net);
    free(m);
}

int orc_test_model(orc_dynmodel* m, const float* x, const float* y, uint32_t n,
                   double thr, float* max_err, float* out_opt)
{
    /* rt-neural-generic.cpp:900-955: gains forced to 1, params forced to 0 */
    float* out = (float*)malloc(sizeof(float) * (n ? n : 1));
    const float ig = m->input_gain, og = m->output_gain;
    m->input_gain = 1.f; m->output_gain = 1.f;
    const float p1 = m->param1Coeff.target, p2 = m->param2Coeff.target;
    orc_linsm_set_target(&m->param1Coeff, 0.f); orc_linsm_clear_to_target(&m->param1Coeff);
    orc_linsm_set_target(&m->param2Coeff, 0.f); orc_linsm_clear_to_target(&m->param2Coeff);
    memcpy(out, x, sizeof(float) * n);
    orc_apply_model(m, out, n);
    m->input_gain = ig; m->output_gain = og;
    orc_linsm_set_target(&m->param1Coeff, p1); orc_linsm_clear_to_target(&m->param1Coeff);
    orc_linsm_set_target(&m->param2Coeff, p2); orc_linsm_clear_to_target(&m->param2Coeff);
    int n_err = 0; float worst = 0.f;
    for (uint32_t i = 0; i < n; ++i) {
        const float e = fabsf(out[i] - y[i]);
        if (e > worst) worst = e;
        if ((double)e > thr) ++n_err;
    }
    if (max_err) *max_err = worst;
    if (out_opt) memcpy(out_opt, out, sizeof(float) * n);
    free(out);
    return n_err;
}

/* ======================================================================
 * Plugin instance mirror
 * ==================================================================== */

#define DEPTH_FREQ 75.0f
#define DEPTH_Q 0.707f
#define PRESENCE_FREQ 900.0f
#define PRESENCE_Q 0.707f

void orc_controls_default(orc_controls* c)
{
    /* lv2:default of ports 4..24, rt-neural-generic.ttl:94-313 */
    c->in_lpf_pc = 66.216f; c->pregain_db = 0.f; c->net_bypass = 0.f;
    c->param1 = 0.f; c->param2 = 0.f; c->eq_bypass = 0.f; c->eq_position = 0.f;
    c->bass_boost_db = 0.f; c->bass_freq = 305.f; c->mid_boost_db = 0.f; c->mid_freq = 750.f;
    c->mid_q = 0.707f; c->mid_type = 0.f; c->treble_boost_db = 0.f; c->treble_freq = 2000.f;
    c->depth_boost_db = 0.f; c->presence_boost_db = 0.f; c->dc_blocker = 1.f;
    c->master_db = 0.f; c->enabled = 1.f;
}

void orc_plugin_init(orc_plugin* p, double sr)
{
    /* instantiate(), rt-neural-generic.cpp:283-321 */
    memset(p, 0, sizeof(*p));
    p->samplerate = sr;
    orc_expsm_init(&p->preGain);
    orc_expsm_set_sample_rate(&p->preGain, (float)sr);
    orc_expsm_set_time_constant(&p->preGain, 0.1f);
    orc_expsm_set_target(&p->preGain, 1.f);
    orc_expsm_clear_to_target(&p->preGain);
    orc_expsm_init(&p->masterGain);
    orc_expsm_set_sample_rate(&p->masterGain, (float)sr);
    orc_expsm_set_time_constant(&p->masterGain, 0.1f);
    orc_expsm_set_target(&p->masterGain, 0.f);
    orc_expsm_clear_to_target(&p->masterGain);

    orc_biquad_init(&p->dc_blocker, ORC_BQ_HIGHPASS, 35.0f / sr, 0.707f, 0.0f);
    p->in_lpf_pc_old = 66.216f;
    orc_biquad_init(&p->in_lpf, ORC_BQ_LOWPASS, orc_lpf_fc(p->in_lpf_pc_old), 0.707f, 0.0f);

    p->bass_boost_db_old = 0.f;  p->bass_freq_old = 250.f;
    orc_biquad_init(&p->bass, ORC_BQ_LOWSHELF, p->bass_freq_old / sr, 0.707f, p->bass_boost_db_old);
    p->mid_boost_db_old = 0.f;   p->mid_freq_old = 600.f; p->mid_q_old = 0.707f; p->mid_type_old = 0.f;
    orc_biquad_init(&p->mid, ORC_BQ_PEAK, p->mid_freq_old / sr, p->mid_q_old, p->mid_boost_db_old);
    p->treble_boost_db_old = 0.f; p->treble_freq_old = 1500.f;
    orc_biquad_init(&p->treble, ORC_BQ_HIGHSHELF, p->treble_freq_old / sr, 0.707f, p->treble_boost_db_old);
    p->depth_boost_db_old = 0.f;
    orc_biquad_init(&p->depth, ORC_BQ_PEAK, DEPTH_FREQ / sr, DEPTH_Q, p->depth_boost_db_old);
    p->presence_boost_db_old = 0.f;
    orc_biquad_init(&p->presence, ORC_BQ_HIGHSHELF, PRESENCE_FREQ / sr, PRESENCE_Q, p->presence_boost_db_old);

    p->loading = 1;
    p->model = NULL;
}

void orc_plugin_activate(orc_plugin* p)
{
    /* rt-neural-generic.cpp:337-351 (model reset is #if 0'd out) */
    orc_expsm_clear_to_target(&p->preGain);
    orc_expsm_clear_to_target(&p->masterGain);
    if (p->model) p->model->paramFirstRun = 1;
}

void orc_plugin_set_model(orc_plugin* p, orc_dynmodel* m)
{
    p->model = m;          /* :872 */
    p->loading = 0;        /* :889 */
}

static void tone_controls(orc_plugin* p, const orc_controls* c, float* buf, uint32_t n)
{
    /* applyToneControls, rt-neural-generic.cpp:50-140 */
    const double sr = p->samplerate;
    int ch;

    ch = 0;
    if (c->bass_boost_db != p->bass_boost_db_old) { p->bass_boost_db_old = c->bass_boost_db; ++ch; }
    if (c->bass_freq != p->bass_freq_old)         { p->bass_freq_old = c->bass_freq; ++ch; }
    if (ch) orc_biquad_set(&p->bass, ORC_BQ_LOWSHELF, c->bass_freq / sr, 0.707f, c->bass_boost_db);

    ch = 0;
    if (c->mid_boost_db != p->mid_boost_db_old) { p->mid_boost_db_old = c->mid_boost_db; ++ch; }
    if (c->mid_freq != p->mid_freq_old)         { p->mid_freq_old = c->mid_freq; ++ch; }
    if (c->mid_q != p->mid_q_old)               { p->mid_q_old = c->mid_q; ++ch; }
    if (c->mid_type != p->mid_type_old)         { p->mid_type_old = c->mid_type; ++ch; }
    if (ch)
        orc_biquad_set(&p->mid, c->mid_type == 1.0f ? ORC_BQ_BANDPASS : ORC_BQ_PEAK,
                       c->mid_freq / sr, c->mid_q, c->mid_boost_db);

    ch = 0;
    if (c->treble_boost_db != p->treble_boost_db_old) { p->treble_boost_db_old = c->treble_boost_db; ++ch; }
    if (c->treble_freq != p->treble_freq_old)         { p->treble_freq_old = c->treble_freq; ++ch; }
    if (ch) orc_biquad_set(&p->treble, ORC_BQ_HIGHSHELF, c->treble_freq / sr, 0.707f, c->treble_boost_db);

    if (c->depth_boost_db != p->depth_boost_db_old) {
        p->depth_boost_db_old = c->depth_boost_db;
        orc_biquad_set(&p->depth, ORC_BQ_PEAK, DEPTH_FREQ / sr, DEPTH_Q, c->depth_boost_db);
    }
    if (c->presence_boost_db != p->presence_boost_db_old) {
        p->presence_boost_db_old = c->presence_boost_db;
        orc_biquad_set(&p->presence, ORC_BQ_HIGHSHELF, PRESENCE_FREQ / sr, PRESENCE_Q, c->presence_boost_db);
    }

    if (c->mid_type == 1.0f) {                    /* BANDPASS: mid only, :130-132 */
        orc_biquad_block(&p->mid, buf, buf, n);
    } else {                                      /* :133-139 */
        orc_biquad_block(&p->depth, buf, buf, n);
        orc_biquad_block(&p->bass, buf, buf, n);
        orc_biquad_block(&p->mid, buf, buf, n);
        orc_biquad_block(&p->treble, buf, buf, n);
        orc_biquad_block(&p->presence, buf, buf, n);
    }
}

void orc_plugin_run(orc_plugin* p, const orc_controls* c, const float* in, float* out, uint32_t n)
{
    /* run(), rt-neural-generic.cpp:489-518 then :607-659 */
    const float pregain = orc_db_co(c->pregain_db);
    const float master = orc_db_co(c->master_db);
    const int net_bypass = c->net_bypass > 0.5f;
    const int enabled = c->enabled > 0.5f;

    orc_expsm_set_target(&p->preGain, pregain);
    if (c->in_lpf_pc != p->in_lpf_pc_old) {
        orc_biquad_set(&p->in_lpf, ORC_BQ_LOWPASS, orc_lpf_fc(c->in_lpf_pc), 0.707f, 0.0f);
        p->in_lpf_pc_old = c->in_lpf_pc;
    }
    if (n == 0) return;
    if (!enabled) {
        if (out != in) memcpy(out, in, sizeof(float) * n);
        return;
    }

    if (c->in_lpf_pc != 0.0f) orc_biquad_block(&p->in_lpf, out, in, n);
    else if (out != in) memcpy(out, in, sizeof(float) * n);
    for (uint32_t i = 0; i < n; ++i) out[i] = out[i] * orc_expsm_next(&p->preGain);   /* applyGainRamp :34-38 */
    if (c->eq_position == 1.0f && c->eq_bypass == 0.0f) tone_controls(p, c, out, n);
    if (p->model && !net_bypass) {
        orc_linsm_set_target(&p->model->param1Coeff, c->param1);
        orc_linsm_set_target(&p->model->param2Coeff, c->param2);
        if (p->model->paramFirstRun) {
            p->model->paramFirstRun = 0;
            orc_linsm_clear_to_target(&p->model->param1Coeff);
            orc_linsm_clear_to_target(&p->model->param2Coeff);
        }
        orc_apply_model(p->model, out, n);
    }
    if (c->dc_blocker == 1.0f) orc_biquad_block(&p->dc_blocker, out, out, n);
    if (c->eq_position == 0.0f && c->eq_bypass == 0.0f) tone_controls(p, c, out, n);
    orc_expsm_set_target(&p->masterGain, p->loading ? 0.f : master);
    for (uint32_t i = 0; i < n; ++i) out[i] = out[i] * orc_expsm_next(&p->masterGain);
}
