// asan_harness.cpp — the host-only half of the library (json parser, loader, weight packers, control-rate DSP design)
// under AddressSanitizer + UndefinedBehaviorSanitizer. CPU test infrastructure: built by `make asan` from the
// product's own sources (aidax_model.cpp + json_min.h, aidax_pack.cpp, aidax_dsp_host.cpp), never shipped, never run
// on the GPU box. Usage: asan_harness <model files...>; every file is loaded (a clean error code is fine), and what
// loads goes through every packer the pool would pick for it; then the biquad designs and the control reduction run
// over a grid of control values. Exit code 0 unless a sanitizer aborts the process.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../aidadsp-lv2_amd/csrc/aidax_internal.h"

namespace aidax {
// the pool's predicate lives next to the HIP code; here every architecture the loader can parse goes on to the packers
bool model_supported(const aidax_model&) { return true; }
}  // namespace aidax

using namespace aidax;

static bool table_width(int h)
{
    for (int w : { 8, 12, 16, 20, 24, 32, 40, 64, 80 })
        if (w == h) return true;
    return false;
}

int main(int argc, char** argv)
{
    int loaded = 0, rejected = 0, packed = 0, refused = 0;
    double sink = 0.0;
    for (int i = 1; i < argc; ++i) {
        aidax_model* m = nullptr;
        const int rc = aidax_model_load(argv[i], &m);
        if (rc != AIDAX_OK || !m) {
            if (!aidax_last_error() || !aidax_last_error()[0]) { std::fprintf(stderr, "%s: error code %d without a message\n", argv[i], rc); return 2; }
            ++rejected;
            continue;
        }
        ++loaded;
        aidax_model_info_t info{};
        (void)aidax_model_info(m, &info);
        std::vector<float> gi(static_cast<size_t>(info.n_golden) + 1), go(static_cast<size_t>(info.n_golden) + 1);
        (void)aidax_model_golden(m, gi.data(), go.data(), static_cast<uint32_t>(info.n_golden));
        auto use = [&](const std::vector<float>& w) { for (float v : w) sink += v; ++packed; };
        try {
            uint32_t st = 0, a = 0, b = 0;
            if (is_conv_model(*m)) {
                bool ok = m->input_size == 1 && m->hidden <= 16 && m->n_rnn <= kMaxConvLayers;
                for (int l = 0; ok && l < m->n_rnn; ++l) ok = m->layers[l].out_size == m->hidden && m->layers[l].ksize <= 8;
                if (ok) { ConvDesc d{}; use(pack_conv(*m, &d, &st)); }
            } else {
                if (m->n_rnn == 1 && table_width(m->hidden)) {
                    const std::vector<float> wp = pack_weights(*m);
                    use(wp);
                    use(pack_quad(*m, &a, &b));
                    // LSTM records that come twice (rotation order for the latency-bound kernels, natural order for k_nn:
                    // aidax_layout.h) must hold the same weights: entry kk of a lane's rotation-order row is the natural
                    // row's entry for unit own + ((pos - kk) & 15) resp. the other row's unit, zeros beyond H
                    if (m->cell == AIDAX_CELL_LSTM && lstm_has_alt_pack(m->hidden)) {
                        const int H = m->hidden;
                        const LaneMap M = lstm_map(H);
                        const int KW = lstm_row_weights(H, true), row_r = KW + kMaxInputs + 1, row_n = H + kMaxInputs + 1;
                        const float* rot = wp.data();
                        const float* nat = wp.data() + lstm_alt_pack_offset(H);
                        if (wp.size() != static_cast<size_t>(lstm_alt_pack_offset(H)) + static_cast<size_t>(lstm_pack_regs(H, false)) * kWave) {
                            std::fprintf(stderr, "%s: pack size\n", argv[i]); return 3;
                        }
                        for (int lane = 0; lane < kWave; ++lane) {
                            const int slot = lane % M.slots, own = 16 * ((slot >> 4) & 1), pos = slot & 15;
                            for (int e = 0; e < M.GPL; ++e)
                                for (int kk = 0; kk < KW; ++kk) {
                                    const int k = kk < 16 ? own + ((pos - kk) & 15) : (16 - own) + (kk - 16);
                                    const float want = k < H ? nat[(e * row_n + k) * kWave + lane] : 0.f;
                                    if (rot[(e * row_r + kk) * kWave + lane] != want) { std::fprintf(stderr, "%s: rotation-order record differs at lane %d row %d entry %d\n", argv[i], lane, e, kk); return 3; }
                                }
                            for (int e = 0; e < M.GPL; ++e)
                                for (int t = 0; t < kMaxInputs + 1; ++t)          // input weights and bias ride along unchanged
                                    if (rot[(e * row_r + KW + t) * kWave + lane] != nat[(e * row_n + H + t) * kWave + lane]) { std::fprintf(stderr, "%s: input / bias entries differ\n", argv[i]); return 3; }
                        }
                    }
                }
                if (mfma_form_fits(*m)) {
                    MfmaDesc d{};
                    const std::vector<float> wp = pack_mfma(*m, &d, &st);
                    use(wp);
                    // k_gru_gs's record: the three bf16 terms of every fragment entry add up to the fp32 weight k_gru_gm
                    // multiplies with — exactly — and shrink by 2^-8 per term (split_bf16x3)
                    if (d.gs_off != 0 && d.gm_off == 0) {
                        // k_lstm_gs's record (one-layer LSTM, unit-major tiles): per wave [4][64] input k-step, [4][64][4] bias quads,
                        // [4 gates][KS2][3 terms][64 lanes][8 bf16] — the terms against the fp32 weight of the tile record k_mfma reads
                        const int H = d.hidden, KS2 = (H + 31) / 32;
                        const size_t per_wave = 4 * kWave + 4 * kWave * 4 + static_cast<size_t>(4) * KS2 * 3 * kWave * 4;
                        auto widen = [](uint32_t half) { const uint32_t u = half << 16; float f; std::memcpy(&f, &u, sizeof f); return f; };
                        for (int w = 0; w < H / 16; ++w)
                            for (int g = 0; g < 4; ++g)
                                for (int ks = 0; ks < KS2; ++ks)
                                    for (int lane = 0; lane < kWave; ++lane)
                                        for (int i = 0; i < 8; ++i) {
                                            const int u = 16 * w + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + i;
                                            float term[3];
                                            for (int t = 0; t < 3; ++t) {
                                                uint32_t pair;
                                                std::memcpy(&pair, &wp[d.gs_off + w * per_wave + 4 * kWave + 4 * kWave * 4 + ((((static_cast<size_t>(g) * KS2 + ks) * 3 + t) * kWave + lane) * 4) + i / 2], sizeof pair);
                                                term[t] = widen((pair >> (16 * (i & 1))) & 0xffffu);
                                            }
                                            const int T = u / 4, r = (u % 4) * 4 + g, kk = k / 4;
                                            const float want = k < H ? wp[d.L[0].w_big_off + ((((static_cast<size_t>(T / d.tpw) * (H / 16) + kk / 4) * kWave + (r | (k & 3) << 4)) * 4 + kk % 4) * d.tpw + T % d.tpw)] : 0.f;
                                            const float sum = (term[2] + term[1]) + term[0];
                                            if (std::isfinite(want) && std::fabs(want) < 1e38f && (sum != want || std::fabs(term[1]) > std::fabs(want) / 256.f + 1e-38f || std::fabs(term[2]) > std::fabs(want) / 65536.f + 1e-38f)) {
                                                std::fprintf(stderr, "%s: LSTM split record differs at wave %d gate %d k %d unit %d: %.9g + %.9g + %.9g vs %.9g\n", argv[i], w, g, k, u, term[0], term[1], term[2], want);
                                                aidax_model_free(m);
                                                return 3;
                                            }
                                        }
                    }
                    if (d.gs_off != 0 && d.gm_off != 0) {
                        const int H = d.hidden, KS = H / 4, KS2 = (H + 31) / 32;
                        auto widen = [](uint32_t half) { const uint32_t u = half << 16; float f; std::memcpy(&f, &u, sizeof f); return f; };
                        for (int w = 0; w < H / 16; ++w)
                            for (int g = 0; g < 3; ++g)
                                for (int ks = 0; ks < KS2; ++ks)
                                    for (int lane = 0; lane < kWave; ++lane)
                                        for (int i = 0; i < 8; ++i) {
                                            const int k = 32 * ks + 8 * (lane >> 4) + i;
                                            float term[3];
                                            for (int t = 0; t < 3; ++t) {
                                                uint32_t pair;
                                                std::memcpy(&pair, &wp[d.gs_off + ((((static_cast<size_t>(w) * 3 + g) * KS2 + ks) * 3 + t) * kWave + lane) * 4 + i / 2], sizeof pair);
                                                term[t] = widen((pair >> (16 * (i & 1))) & 0xffffu);
                                            }
                                            // the gate-major fp32 record: [3 input rows][KS k-steps][3 gates][64 lanes], lane = row | (k & 3) << 4
                                            const float want = k < H ? wp[d.gm_off + static_cast<size_t>(w) * kWave * (3 + 3 * KS + 16) + (3 + 3 * (k / 4) + g) * kWave + ((k & 3) << 4 | (lane & 15))] : 0.f;
                                            const float sum = (term[2] + term[1]) + term[0];
                                            if (std::isfinite(want) && std::fabs(want) < 1e38f && (sum != want || std::fabs(term[1]) > std::fabs(want) / 256.f + 1e-38f || std::fabs(term[2]) > std::fabs(want) / 65536.f + 1e-38f)) {
                                                std::fprintf(stderr, "%s: split record differs at wave %d gate %d k %d row %d: %.9g + %.9g + %.9g vs %.9g\n", argv[i], w, g, k, lane & 15, term[0], term[1], term[2], want);
                                                return 3;
                                            }
                                        }
                    }
                }
                if (is_stack_model(*m) && m->n_rnn <= kMaxStackLayers && m->hidden <= 128 && m->hidden % 4 == 0) { StackDesc d{}; use(pack_stack(*m, &d, &st)); }
            }
        } catch (const std::exception&) {
            ++refused;                                      // a packer may refuse a shape; it must not corrupt memory doing so
        }
        aidax_model_free(m);
    }
    // control-rate host code: every filter type over a grid, and the per-stream record for a sweep of control values
    for (int type = 0; type < 7; ++type)
        for (double fc : { 0.0007, 0.01, 0.125, 0.3, 0.495 })
            for (double q : { 0.2, 0.707, 5.0 })
                for (double g : { -20.0, -0.5, 0.0, 6.0, 20.0 }) {
                    double c[5];
                    design_biquad(type, fc, q, g, c);
                    sink += c[0] + c[4];
                }
    aidax_controls c{};
    aidax_controls_default(&c);
    const float sweep[] = { -200.f, -96.f, -1.f, 0.f, 0.5f, 1.f, 25.f, 100.f, 20000.f };
    float* fields = reinterpret_cast<float*>(&c);
    const size_t nf = sizeof(aidax_controls) / sizeof(float);
    for (size_t f = 0; f < nf; ++f)
        for (float v : sweep) {
            aidax_controls k = c;
            reinterpret_cast<float*>(&k)[f] = v;
            for (double sr : { 8000.0, 44100.0, 48000.0, 192000.0 }) {
                StreamCtl o{};
                build_stream_ctl(k, sr, (f & 1) != 0, (f & 2) != 0, exp_smoother_coef(static_cast<float>(sr), 0.1f), 0.1f * 48000.f, &o);
                sink += o.pre_target + o.bq[0][0];
            }
        }
    (void)fields;
    std::printf("asan_harness: %d loaded, %d rejected, %d packs, %d refused (sink %g)\n", loaded, rejected, packed, refused, sink);
    return 0;
}
