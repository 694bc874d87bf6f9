// CPU check of pack_mfma's fragment layout: decode the fragments the way k_mfma's MFMAs consume them and
// compare the gate pre-activations of one frame with the plain Keras-layout computation.
#include <cmath>
#include <cstdio>
#include <vector>
#include "../aidadsp-lv2_amd/csrc/aidax_internal.h"
using namespace aidax;
int main(int argc, char** argv)
{
    aidax_model* m = nullptr;
    if (aidax_model_load(argv[1], &m) != 0) { printf("load failed: %s\n", aidax_last_error()); return 1; }
    MfmaDesc d; uint32_t st;
    std::vector<float> W = pack_mfma(*m, &d, &st);
    const int H = d.hidden, Ht = d.hidden_true, NW = d.waves, TPW = d.tpw;
    printf("H=%d Ht=%d NW=%d TPW=%d layers=%d state=%u\n", H, Ht, NW, TPW, d.n_layers, st);
    double worst = 0;
    std::vector<float> below;
    for (int l = 0; l < d.n_layers; ++l) {
        const Layer& L = m->layers[l];
        const MfmaLayer& M = d.L[l];
        const int I = L.in_size, G = L.type == Layer::LSTM ? 4 : 3, R = G * Ht;
        std::vector<float> xin(l == 0 ? 4 : H, 0.f), h(H, 0.f);
        for (int i = 0; i < I; ++i) xin[i] = 0.3f + 0.1f * i - 0.05f * l;
        for (int i = 0; i < Ht; ++i) h[i] = 0.01f * ((i * 37) % 19 - 9);
        const int g_in = l == 0 ? 0 : H / 16, g_tot = g_in + H / 16;
        for (int w = 0; w < NW; ++w)
            for (int tl = 0; tl < TPW; ++tl)
                for (int r = 0; r < 16; ++r) {
                    const int T = w * TPW + tl, u = 4 * T + (r >> 2), g = r & 3;
                    double acc = W[M.b_off + u * 4 + g];
                    if (l == 0)
                        for (int q = 0; q < 4; ++q) acc += (double)W[M.w_in_off + ((size_t)w * 64 + (q * 16 + r)) * TPW + tl] * xin[q];
                    for (int grp = 0; grp < g_tot; ++grp)
                        for (int j = 0; j < 4; ++j)
                            for (int q = 0; q < 4; ++q) {
                                const int lane = q * 16 + r;
                                const float a = W[M.w_big_off + (((size_t)w * g_tot + grp) * 64 + lane) * 4 * TPW + j * TPW + tl];
                                const int kloc = 16 * (grp < g_in ? grp : grp - g_in) + 4 * j + q;
                                acc += (double)a * (grp < g_in ? xin[kloc] : h[kloc]);
                            }
                    // reference
                    double ref = 0;
                    if (u < Ht) {
                        const float* Wk = L.w0.data(); const float* U = L.w1.data(); const float* b = L.w2.data();
                        if (G == 4) {
                            const int col = g * Ht + u;
                            ref = b[col];
                            for (int k = 0; k < I; ++k) ref += (double)Wk[(size_t)k * R + col] * xin[k];
                            for (int k = 0; k < Ht; ++k) ref += (double)U[(size_t)k * R + col] * h[k];
                        } else {
                            const int col = (g == 0 ? 0 : g == 1 ? Ht : 2 * Ht) + u;
                            if (g <= 1) { ref = b[col] + b[R + col]; for (int k = 0; k < I; ++k) ref += (double)Wk[(size_t)k * R + col] * xin[k]; for (int k = 0; k < Ht; ++k) ref += (double)U[(size_t)k * R + col] * h[k]; }
                            else if (g == 2) { ref = b[R + col]; for (int k = 0; k < Ht; ++k) ref += (double)U[(size_t)k * R + col] * h[k]; }
                            else { ref = b[col]; for (int k = 0; k < I; ++k) ref += (double)Wk[(size_t)k * R + col] * xin[k]; }
                        }
                    }
                    const double e = std::fabs(acc - ref);
                    if (e > worst) { worst = e; }
                    if (e > 1e-5) { printf("layer %d unit %d gate %d: got %g want %g\n", l, u, g, acc, ref); }
                }
    }
    printf("worst |gate error| = %g\n", worst);
    return 0;
}
