// aidax_dsp_host.cpp — control-rate DSP that stays on the host: biquad design,
// dB -> linear, the ANTIALIASING map, smoother coefficients, and the reduction
// of the 20 control ports to the per-stream record the kernels read.
//
// Built with -ffp-contract=off: the designs must round exactly like the
// reference's calcBiquad (common/Biquad.cpp:67-165) so that the fp64 biquad
// kernels are bit-exact against it. Each filter family is written as
// "denominator polynomial, numerator polynomial, normalise", keeping the
// reference's association order inside every sum and product.
#include <cmath>

#include "aidax_internal.h"
#include <cstring>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace aidax {

namespace {

// (1 + p*K + KK) and (1 - p*K + KK): the two quadratic forms every design uses,
// with the K-coefficient p already multiplied in by the caller where the
// reference multiplies first (p*K evaluated as written there).
inline double quad_plus(double lead, double pk, double kk) { return lead + pk + kk; }
inline double quad_minus(double lead, double pk, double kk) { return lead - pk + kk; }

}  // namespace

void design_biquad(int type, double fc, double q, double gain_db, double c[5])
{
    const double V = std::pow(10, std::fabs(gain_db) / 20.0);
    const double K = std::tan(M_PI * fc);
    const double KK = K * K;
    const bool boost = gain_db >= 0;
    double a0 = 1.0, a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;

    if (type >= 0 && type <= 3) {
        // all-pole part shared by lowpass/highpass/bandpass/notch (Biquad.cpp:72-106)
        const double koq = K / q;
        const double norm = 1 / quad_plus(1, koq, KK);
        b1 = 2 * (KK - 1) * norm;
        b2 = quad_minus(1, koq, KK) * norm;
        switch (type) {
        case 0: a0 = KK * norm;        a1 = 2 * a0;  a2 = a0;  break;
        case 1: a0 = 1 * norm;         a1 = -2 * a0; a2 = a0;  break;
        case 2: a0 = koq * norm;       a1 = 0;       a2 = -a0; break;
        default: a0 = (1 + KK) * norm; a1 = b1;      a2 = a0;  break;   // notch: a1 == b1 term for term
        }
    } else if (type == 4) {
        // peaking EQ (Biquad.cpp:108-125): V rides on the numerator (boost) or denominator (cut)
        const double plain = 1 / q * K, gained = V / q * K;
        const double num_k = boost ? gained : plain;
        const double den_k = boost ? plain : gained;
        const double norm = 1 / quad_plus(1, den_k, KK);
        a0 = quad_plus(1, num_k, KK) * norm;
        a1 = 2 * (KK - 1) * norm;
        a2 = quad_minus(1, num_k, KK) * norm;
        b1 = a1;
        b2 = quad_minus(1, den_k, KK) * norm;
    } else if (type == 5) {
        // low shelf (Biquad.cpp:126-143): the gained side uses sqrt(2V)*K and V*K*K
        const double r2k = std::sqrt(2) * K, s2vk = std::sqrt(2 * V) * K, vkk = V * K * K;
        if (boost) {
            const double norm = 1 / quad_plus(1, r2k, KK);
            a0 = quad_plus(1, s2vk, vkk) * norm;
            a1 = 2 * (vkk - 1) * norm;
            a2 = quad_minus(1, s2vk, vkk) * norm;
            b1 = 2 * (KK - 1) * norm;
            b2 = quad_minus(1, r2k, KK) * norm;
        } else {
            const double norm = 1 / quad_plus(1, s2vk, vkk);
            a0 = quad_plus(1, r2k, KK) * norm;
            a1 = 2 * (KK - 1) * norm;
            a2 = quad_minus(1, r2k, KK) * norm;
            b1 = 2 * (vkk - 1) * norm;
            b2 = quad_minus(1, s2vk, vkk) * norm;
        }
    } else if (type == 6) {
        // high shelf (Biquad.cpp:144-161): the gained side leads with V instead of 1
        const double r2k = std::sqrt(2) * K, s2vk = std::sqrt(2 * V) * K;
        if (boost) {
            const double norm = 1 / quad_plus(1, r2k, KK);
            a0 = quad_plus(V, s2vk, KK) * norm;
            a1 = 2 * (KK - V) * norm;
            a2 = quad_minus(V, s2vk, KK) * norm;
            b1 = 2 * (KK - 1) * norm;
            b2 = quad_minus(1, r2k, KK) * norm;
        } else {
            const double norm = 1 / quad_plus(V, s2vk, KK);
            a0 = quad_plus(1, r2k, KK) * norm;
            a1 = 2 * (KK - 1) * norm;
            a2 = quad_minus(1, r2k, KK) * norm;
            b1 = 2 * (KK - V) * norm;
            b2 = quad_minus(V, s2vk, KK) * norm;
        }
    }
    c[0] = a0; c[1] = a1; c[2] = a2; c[3] = b1; c[4] = b2;
}

float db_to_coeff(float db)
{
    // DB_CO, rt-neural-generic.h:160
    return db > -90.0f ? powf(10.0f, db * 0.05f) : 0.0f;
}

float lpf_fc(float percent)
{
    // MAP(pc, 0, 100, INLPF_MAX_CO, INLPF_MIN_CO), float arithmetic (rt-neural-generic.h:167,178-179)
    const float hi = 0.99f * 0.5f, lo = 0.25f * 0.5f;
    return ((percent - 0.0f) * (lo - hi) / (100.0f - 0.0f)) + hi;
}

float exp_smoother_coef(float samplerate, float t60)
{
    // ExponentialValueSmoother::setTimeConstant + updateCoef (ValueSmoother.hpp:106-115, :147-151)
    const float tau = t60 * (float)(1.0 / 6.91);
    return std::exp(-1.f / (tau * samplerate));
}

void build_stream_ctl(const aidax_controls& c, double sr, bool has_model, bool loading,
                      float gain_coef, float p_den, StreamCtl* o)
{
    // Coefficients are a pure function of the current port values: the reference's
    // *_old change detection (rt-neural-generic.cpp:68-127, :514-517) only decides
    // WHEN setBiquad runs, and setBiquad never touches z1/z2 (Biquad.cpp:60-65).
    design_biquad(0, lpf_fc(c.in_lpf_pc), 0.707f, 0.0f, o->bq[BQ_LPF]);                    // :515
    design_biquad(1, 35.0f / sr, 0.707f, 0.0f, o->bq[BQ_DC]);                               // :293
    design_biquad(4, 75.0f / sr, 0.707f, c.depth_boost_db, o->bq[BQ_DEPTH]);                // :122
    design_biquad(5, c.bass_freq / sr, 0.707f, c.bass_boost_db, o->bq[BQ_BASS]);            // :77
    design_biquad(c.mid_type == 1.0f ? 2 : 4, c.mid_freq / sr, c.mid_q, c.mid_boost_db, o->bq[BQ_MID]);   // :98-103
    design_biquad(6, c.treble_freq / sr, 0.707f, c.treble_boost_db, o->bq[BQ_TREBLE]);      // :116
    design_biquad(6, 900.0f / sr, 0.707f, c.presence_boost_db, o->bq[BQ_PRESENCE]);         // :126

    o->pre_target = db_to_coeff(c.pregain_db);                                              // :489
    o->master_target = loading ? 0.f : db_to_coeff(c.master_db);                            // :490, :654
    o->pre_coef = gain_coef;
    o->master_coef = gain_coef;
    o->p_target[0] = c.param1;
    o->p_target[1] = c.param2;
    o->p_den = p_den;
    uint32_t f = 0;
    if (c.enabled > 0.5f) f |= CTL_ENABLED;                                                 // :495
    if (c.in_lpf_pc != 0.0f) f |= CTL_LPF_ON;                                               // :622
    if (c.eq_bypass == 0.0f && c.eq_position == 1.0f) f |= CTL_EQ_PRE;                      // :628
    if (c.eq_bypass == 0.0f && c.eq_position == 0.0f) f |= CTL_EQ_POST;                     // :651
    if (c.mid_type == 1.0f) f |= CTL_EQ_BANDPASS;                                           // :130
    if (has_model && !(c.net_bypass > 0.5f)) f |= CTL_NET_ON;                               // :631-632
    if (c.dc_blocker == 1.0f) f |= CTL_DC_ON;                                               // :646
    o->flags = f;
    o->pad[0] = o->pad[1] = 0;
}

}  // namespace aidax

using namespace aidax;

extern "C" {

AIDAX_API void aidax_controls_default(aidax_controls* c)
{
    if (!c) return;
    // lv2:default of ports 4..24 (rt-neural-generic.ttl:94-313)
    c->in_lpf_pc = 66.216f; c->pregain_db = 0.f; c->net_bypass = 0.f; c->param1 = 0.f; c->param2 = 0.f;
    c->eq_bypass = 0.f; c->eq_position = 0.f; c->bass_boost_db = 0.f; c->bass_freq = 305.f;
    c->mid_boost_db = 0.f; c->mid_freq = 750.f; c->mid_q = 0.707f; c->mid_type = 0.f;
    c->treble_boost_db = 0.f; c->treble_freq = 2000.f; c->depth_boost_db = 0.f; c->presence_boost_db = 0.f;
    c->dc_blocker = 1.f; c->master_db = 0.f; c->enabled = 1.f;
}

AIDAX_API int aidax_biquad_design(int type, double fc, double q, double gain_db, double coeffs[5])
{
    if (!coeffs || type < 0 || type > 6) return fail(AIDAX_ERR_ARG, "biquad type out of range");
    design_biquad(type, fc, q, gain_db, coeffs);
    return AIDAX_OK;
}

AIDAX_API float aidax_db_to_coeff(float db) { return db_to_coeff(db); }
AIDAX_API float aidax_lpf_fc(float percent) { return lpf_fc(percent); }

// Placement rule of include/aidax.h (pure: the device count and the loads come from the caller)
AIDAX_API int aidax_pick_device(const char* spec, int device_count, const uint32_t* load, int* device_out)
{
    if (!device_out || device_count <= 0) return fail(AIDAX_ERR_ARG, "aidax_pick_device: no device to choose from");
    bool cand[256] = {};
    const int n = device_count < 256 ? device_count : 256;
    if (!spec || !*spec) cand[0] = true;
    else if (!std::strcmp(spec, "auto")) { for (int i = 0; i < n; ++i) cand[i] = true; }
    else {
        const char* p = spec;
        while (*p) {
            if (*p < '0' || *p > '9') return fail(AIDAX_ERR_ARG, "aidax_pick_device: device list is not 'auto' or indices / ranges like 0-3,6");
            long a = 0, b;
            while (*p >= '0' && *p <= '9') { a = a * 10 + (*p++ - '0'); if (a > 100000) return AIDAX_ERR_ARG; }
            b = a;
            if (*p == '-') {
                ++p;
                if (*p < '0' || *p > '9') return AIDAX_ERR_ARG;
                b = 0;
                while (*p >= '0' && *p <= '9') { b = b * 10 + (*p++ - '0'); if (b > 100000) return AIDAX_ERR_ARG; }
                if (b < a) return AIDAX_ERR_ARG;
            }
            for (long i = a; i <= b && i < n; ++i) cand[i] = true;
            if (*p == ',') { ++p; if (!*p) return AIDAX_ERR_ARG; }
            else if (*p) return AIDAX_ERR_ARG;
        }
    }
    int best = -1;
    for (int i = 0; i < n; ++i)
        if (cand[i] && (best < 0 || (load ? load[i] : 0u) < (load ? load[best] : 0u))) best = i;
    if (best < 0) return fail(AIDAX_ERR_ARG, "aidax_pick_device: the device list names no device of this machine");
    *device_out = best;
    return AIDAX_OK;
}

AIDAX_API int aidax_pick_hub(const int* hub_device, const uint32_t* hub_free_seats, int n_hubs, int current_device,
                             const char* spec, int device_count, const uint32_t* load, int* index_out, int* device_out)
{
    if (!index_out || !device_out || n_hubs < 0 || (n_hubs > 0 && (!hub_device || !hub_free_seats))) return aidax::fail(AIDAX_ERR_ARG, "pick_hub: null argument");
    if (current_device >= device_count) return aidax::fail(AIDAX_ERR_ARG, "pick_hub: current device out of range");
    *index_out = -1;
    if (current_device >= 0) {
        // a playing instance keeps its device: its DSP state moves seat to seat on the device (aidax_hub_adopt)
        for (int i = 0; i < n_hubs; ++i)
            if (hub_device[i] == current_device && hub_free_seats[i] > 0) { *index_out = i; break; }
        *device_out = current_device;
        return AIDAX_OK;
    }
    for (int i = 0; i < n_hubs; ++i)
        if (hub_free_seats[i] > 0) { *index_out = i; *device_out = hub_device[i]; return AIDAX_OK; }
    return aidax_pick_device(spec, device_count, load, device_out);
}

}  // extern "C"
