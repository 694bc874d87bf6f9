/*
 * ref_shim.cpp — extern "C" handles onto the REFERENCE's own Biquad and value
 * smoother classes, compiled from the sources where they lie:
 *     /root/reference/common/Biquad.cpp, Biquad.h, ValueSmoother.hpp
 * (see oracle/Makefile, target _ref). TEST INFRASTRUCTURE ONLY. This file holds
 * no reference code — it only instantiates the reference classes so the
 * restatement in aidax_oracle_dsp.c can be compared with them bit-for-bit and
 * so tests/golden/make_golden.py can dump fixtures. The resulting
 * oracle/_ref/libaidadsp_ref.so is git-ignored.
 *
 * The reference's RTNeural/LV2-dependent sources (rt-neural-generic.cpp) are
 * unbuildable here (no RTNeural, no LV2 headers) and are not part of _ref.
 */
#include <Biquad.h>
#include <ValueSmoother.hpp>

#include <cstdint>

namespace {
struct BiquadPeek : public Biquad {
    BiquadPeek(int t, double fc, double q, double g) : Biquad(t, fc, q, g) {}
    void coeffs(double* o) const { o[0] = a0; o[1] = a1; o[2] = a2; o[3] = b1; o[4] = b2; }
    void state(double* o) const { o[0] = z1; o[1] = z2; }
};
}

extern "C" {

void* ref_biquad_new(int type, double fc, double q, double gain_db) { return new BiquadPeek(type, fc, q, gain_db); }
void  ref_biquad_free(void* h) { delete static_cast<BiquadPeek*>(h); }
void  ref_biquad_set(void* h, int type, double fc, double q, double gain_db) { static_cast<BiquadPeek*>(h)->setBiquad(type, fc, q, gain_db); }
void  ref_biquad_coeffs(void* h, double* out5) { static_cast<BiquadPeek*>(h)->coeffs(out5); }
void  ref_biquad_state(void* h, double* out2) { static_cast<BiquadPeek*>(h)->state(out2); }
void  ref_biquad_block(void* h, float* out, const float* in, uint32_t n)
{
    BiquadPeek* f = static_cast<BiquadPeek*>(h);
    for (uint32_t i = 0; i < n; ++i) out[i] = f->process(in[i]);
}

void* ref_expsm_new(float sr, float t60, float target)
{
    ExponentialValueSmoother* s = new ExponentialValueSmoother();
    s->setSampleRate(sr); s->setTimeConstant(t60); s->setTargetValue(target); s->clearToTargetValue();
    return s;
}
void  ref_expsm_free(void* h) { delete static_cast<ExponentialValueSmoother*>(h); }
void  ref_expsm_set_target(void* h, float t) { static_cast<ExponentialValueSmoother*>(h)->setTargetValue(t); }
void  ref_expsm_clear(void* h) { static_cast<ExponentialValueSmoother*>(h)->clearToTargetValue(); }
void  ref_expsm_run(void* h, float* out, uint32_t n)
{
    ExponentialValueSmoother* s = static_cast<ExponentialValueSmoother*>(h);
    for (uint32_t i = 0; i < n; ++i) out[i] = s->next();
}

void* ref_linsm_new(float sr, float tau, float target)
{
    LinearValueSmoother* s = new LinearValueSmoother();
    s->setSampleRate(sr); s->setTimeConstant(tau); s->setTargetValue(target); s->clearToTargetValue();
    return s;
}
void  ref_linsm_free(void* h) { delete static_cast<LinearValueSmoother*>(h); }
void  ref_linsm_set_target(void* h, float t) { static_cast<LinearValueSmoother*>(h)->setTargetValue(t); }
void  ref_linsm_clear(void* h) { static_cast<LinearValueSmoother*>(h)->clearToTargetValue(); }
void  ref_linsm_run(void* h, float* out, uint32_t n)
{
    LinearValueSmoother* s = static_cast<LinearValueSmoother*>(h);
    for (uint32_t i = 0; i < n; ++i) out[i] = s->next();
}

}  // extern "C"
