// aidax_pack.cpp — shuffles a model's Keras-layout weights into the register
// order of the kernel's lane mapping (aidax_layout.h): record r of lane l is at
// wpack[r*64 + l], so the kernel fills each weight VGPR with one coalesced read.
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "aidax_internal.h"

namespace aidax {

namespace {

struct Packer {
    std::vector<float> out;
    int regs;
    explicit Packer(int regs_) : out(static_cast<size_t>(regs_) * kWave, 0.f), regs(regs_) {}
    void put(int r, int lane, float v)
    {
        if (r >= regs) throw std::logic_error("pack overflow");
        out[static_cast<size_t>(r) * kWave + lane] = v;
    }
};

// The activations' input scales are folded into the gate rows (weights and bias), so the kernels' recurrence carries no
// multiply for them (LstmCell / GruCell, aidax_kernels.hip): a sigmoid evaluated as 1 / (1 + 2^v) takes its row times
// -log2(e); a sigmoid that shares its instruction stream with a tanh row (S >= 2) is evaluated as 0.5 tanh(0.5 v) + 0.5 and
// takes its row times 0.5, which is exact; tanh rows stay as they are.
constexpr float kNegLog2e = -1.44269504088896340736f;

static std::vector<float> pack_lstm_records(const aidax_model& m, bool rotation_order);

// [lane records | Dense weights + bias in natural order (the pipeline's output wave)] and, for LSTM-32, the lane records
// once more with the recurrent weights in natural order (lstm_has_alt_pack, aidax_layout.h)
std::vector<float> pack_lstm(const aidax_model& m)
{
    const int H = m.layers[0].out_size;
    std::vector<float> out = pack_lstm_records(m, lstm_has_alt_pack(H));
    if (lstm_has_alt_pack(H)) {
        std::vector<float> alt = pack_lstm_records(m, false);
        alt.resize(static_cast<size_t>(lstm_pack_regs(H, false)) * kWave);   // without its Dense tail
        if (static_cast<int>(out.size()) != lstm_alt_pack_offset(H)) throw std::logic_error("pack_lstm: record size");
        out.insert(out.end(), alt.begin(), alt.end());
    }
    return out;
}

static std::vector<float> pack_lstm_records(const aidax_model& m, bool rotation_order)
{
    const Layer& L = m.layers[0];
    const Layer& D = m.layers[1];
    const int H = L.out_size, I = L.in_size, G = 4 * H;
    const LaneMap M = lstm_map(H);
    Packer p(lstm_pack_regs(H, rotation_order));
    const int KW = lstm_row_weights(H, rotation_order);
    auto row_scale = [&](int part, int e) {
        const int gate = part + M.S * e;                            // i, f, g (the candidate, tanh), o
        if (M.S == 1) return gate == 2 ? 1.f : kNegLog2e;
        if (M.S == 2 && e == 0) return kNegLog2e;                   // (i | f): a sigmoid in both halves
        return gate == 2 ? 1.f : 0.5f;                              // rows evaluated in the common tanh form
    };
    const int state_part = M.S == 2 ? 1 : 0;                        // the lanes that keep c and h (LstmCell::kStatePart)
    for (int lane = 0; lane < kWave; ++lane) {
        const int part = lane / M.slots, slot = lane % M.slots;
        int r = 0;
        for (int mm = 0; mm < M.NU; ++mm) {
            const int j = slot + mm * M.slots;
            const bool live = j < H;
            for (int e = 0; e < M.GPL; ++e) {
                const int col = (part + M.S * e) * H + j;           // gate-major column of the json matrices
                const float sc = row_scale(part, e);
                for (int kk = 0; kk < KW; ++kk) {
                    // rotation order: a lane's recurrent weights in the order the kernel uses them — the 16 units of its own
                    // 16-lane row by rotation (row_ror:n delivers the lane n places to the left, cyclically), then (LSTM-32)
                    // the other row's 16 in natural order. S = 4: a row is all the units; those beyond H weigh nothing.
                    int k = kk;
                    if (rotation_order) {
                        const int own = 16 * ((slot >> 4) & 1), pos = slot & 15;
                        k = kk < 16 ? own + ((pos - kk) & 15) : (16 - own) + (kk - 16);
                    }
                    p.put(r++, lane, (live && k < H) ? sc * L.w1[static_cast<size_t>(k) * G + col] : 0.f);
                }
                for (int i = 0; i < kMaxInputs; ++i) p.put(r++, lane, (live && i < I) ? sc * L.w0[static_cast<size_t>(i) * G + col] : 0.f);
                p.put(r++, lane, live ? sc * L.w2[col] : 0.f);
            }
        }
        for (int mm = 0; mm < M.NU; ++mm) {
            const int j = slot + mm * M.slots;
            p.put(r++, lane, (part == state_part && j < H) ? D.w0[j] : 0.f);
        }
        p.put(r++, lane, D.w1[0]);
    }
    for (int j = 0; j < H; ++j) p.out.push_back(D.w0[j]);      // natural-order Dense weights + bias for the
    p.out.push_back(D.w1[0]);                                    // pipeline's output wave (after the lane records)
    return std::move(p.out);
}

std::vector<float> pack_gru(const aidax_model& m)
{
    const Layer& L = m.layers[0];
    const Layer& D = m.layers[1];
    const int H = L.out_size, I = L.in_size, G = 3 * H;
    const LaneMap M = gru_map(H);
    const int KS = H / M.S;
    Packer p(gru_pack_regs(H));
    const float* b0 = L.w2.data();          // input-side bias
    const float* b1 = L.w2.data() + G;      // recurrent-side bias
    for (int lane = 0; lane < kWave; ++lane) {
        const int part = lane / M.slots, slot = lane % M.slots;
        int r = 0;
        for (int mm = 0; mm < M.NU; ++mm) {
            const int j = slot + mm * M.slots;
            const bool live = j < H;
            for (int e = 0; e < 3; ++e) {
                const int col = e * H + j;
                // z, r: sigmoids as 1 / (1 + 2^v); candidate: tanh as 1 - 2 / (1 + 2^v) (tanh_exp_pre; r * (U h + b1) + (W x + b0)
                // is linear in the candidate rows, so the factor goes into all of them)
                const float sc = e < 2 ? kNegLog2e : kTwoLog2e;
                // this lane's K slice of the recurrent row
                for (int k = 0; k < KS; ++k) p.put(r++, lane, live ? sc * L.w1[static_cast<size_t>(part * KS + k) * G + col] : 0.f);
                // z, r: input weights and (b0+b1) ride on part 0 and reach the others through the partial-sum
                // exchange; candidate: every part keeps the input side whole (it is not summed), b1 sits under r*( )
                const bool owns_input = e == 2 || part == 0;
                for (int i = 0; i < kMaxInputs; ++i) p.put(r++, lane, (live && owns_input && i < I) ? sc * L.w0[static_cast<size_t>(i) * G + col] : 0.f);
                p.put(r++, lane, (live && owns_input) ? (e < 2 ? sc * (b0[col] + b1[col]) : sc * b0[col]) : 0.f);
            }
        }
        for (int mm = 0; mm < M.NU; ++mm) {
            const int j = slot + mm * M.slots;
            p.put(r++, lane, (j < H && part == 0) ? kTwoLog2e * b1[2 * H + j] : 0.f);
        }
        for (int mm = 0; mm < M.NU; ++mm) {
            const int j = slot + mm * M.slots;
            p.put(r++, lane, (j < H && part == 0) ? D.w0[j] : 0.f);
        }
        p.put(r++, lane, D.w1[0]);
    }
    for (int j = 0; j < H; ++j) p.out.push_back(D.w0[j]);
    p.out.push_back(D.w1[0]);
    return std::move(p.out);
}

}  // namespace

bool is_stack_model(const aidax_model& m)
{
    if (m.cell == AIDAX_CELL_CONV) return false;
    if (m.n_rnn >= 2) return true;
    // one recurrent layer wider than the widest of variant/generate_variant_hpp.py:4-6 (no register-resident
    // kernel); narrower sizes outside that list stay rejected like the reference rejects them
    return m.n_rnn == 1 && m.hidden > 80;
}

// k_mfma serves recurrent layers of one width <= 128; widths that are not a multiple of 16 are padded with
// zero rows and columns (a padded unit has i = o = 1/2, g = 0: its c and h stay exactly 0).
bool mfma_form_fits(const aidax_model& m)
{
    if (m.cell == AIDAX_CELL_CONV || m.n_rnn < 1 || m.n_rnn > kMaxStackLayers || m.hidden > 128) return false;
    for (int l = 0; l < m.n_rnn; ++l)
        if (m.layers[l].out_size != m.hidden) return false;
    return true;
}

// x = t[0] + t[1] + t[2] EXACTLY, each term a bf16 (bit pattern returned): t[0] = x rounded to nearest even, t[1] the
// same of the remainder, t[2] the remainder of that — 8 + 8 + 8 significant bits, and a remainder of a round-to-nearest
// is at most half an ulp, so |t[1]| <= 2^-8 |x| and |t[2]| <= 2^-16 |x|. (The kernels split h(t-1) the same way on the
// device: v_cvt_pk_bf16_f32 rounds to nearest even.)
void split_bf16x3(float x, uint16_t (&t)[3])
{
    auto rne = [](float v) -> uint16_t {
        uint32_t u;
        std::memcpy(&u, &v, sizeof u);
        if ((u & 0x7f800000u) == 0x7f800000u) return static_cast<uint16_t>(u >> 16);      // inf / nan as they are
        const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
        return static_cast<uint16_t>(((r & 0x7f800000u) == 0x7f800000u ? u : r) >> 16);     // (never round a finite value up to inf)
    };
    auto widen = [](uint16_t b) -> float {
        const uint32_t u = static_cast<uint32_t>(b) << 16;
        float f;
        std::memcpy(&f, &u, sizeof f);
        return f;
    };
    float r = x;
    for (int i = 0; i < 3; ++i) {
        t[i] = rne(r);
        r = std::isfinite(r) ? r - widen(t[i]) : 0.f;                                      // exact (a file's inf / nan stays in the first term)
    }
}

// Weight of gate row `g` of unit `u` of layer `l` against column k of segment `seg` (0 in the padding), as the matrix-core
// kernels multiply it. The sigmoid rows (LSTM i, f, o; GRU z, r) carry the factor -log2(e) in weights and bias, like the
// table kernels' records: the cell update evaluates 1 / (1 + 2^v) without a multiply (sigmoid_pre, aidax_device.h); a GRU's
// candidate rows (both halves) carry 2 log2(e): tanh as 1 - 2 / (1 + 2^v), see tanh_exp_pre.
enum Seg { SEG_IN, SEG_REC, SEG_BIAS };
static float mfma_weight(const aidax_model& m, int l, Seg seg, int u, int g, int k)
{
    const Layer& L = m.layers[l];
    const int Ht = m.hidden;
    const bool lstm = L.type == Layer::LSTM;
    const int G = lstm ? 4 : 3, R = G * Ht, I = L.in_size;
    auto weight_raw = [&]() -> float {
        if (u >= Ht || (seg == SEG_REC && k >= Ht) || (seg == SEG_IN && k >= I)) return 0.f;
        const int H = Ht;                        // Keras column blocks are Ht wide
        const float* W = L.w0.data();      // [I][R]
        const float* U = L.w1.data();      // [H][R]
        const float* b = L.w2.data();      // LSTM [R]; GRU [2][R]
        if (lstm) {
            const int col = g * H + u;
            if (seg == SEG_IN) return k < I ? W[(size_t)k * R + col] : 0.f;
            if (seg == SEG_REC) return U[(size_t)k * R + col];
            return k == 0 ? b[col] : 0.f;
        }
        // GRU rows: 0 = z, 1 = r, 2 = recurrent half of the candidate, 3 = input half of the candidate
        const int col = (g == 0 ? 0 : g == 1 ? H : 2 * H) + u;
        if (seg == SEG_IN) return (g != 2 && k < I) ? W[(size_t)k * R + col] : 0.f;
        if (seg == SEG_REC) return g != 3 ? U[(size_t)k * R + col] : 0.f;
        if (k != 0) return 0.f;
        if (g <= 1) return b[col] + b[R + col];
        return g == 2 ? b[R + col] : b[col];
    };
    const bool sigmoid_row = lstm ? g != 2 : g <= 1;
    const float w = weight_raw();
    return sigmoid_row ? kNegLog2e * w : !lstm ? kTwoLog2e * w : w;
}

// A fragments of v_mfma_f32_16x16x4_f32 for k_mfma, layout in aidax_layout.h (MfmaLayer).
std::vector<float> pack_mfma(const aidax_model& m, MfmaDesc* d, uint32_t* state_floats)
{
    std::vector<float> out;
    *d = MfmaDesc{};
    const int Ht = m.hidden, H = (Ht + 15) & ~15, NW = mfma_waves(H), TPW = H / 4 / NW;
    d->n_layers = m.n_rnn;
    d->hidden = H;
    d->hidden_true = Ht;
    d->tpw = TPW;
    d->waves = NW;
    uint32_t st = 0;
    for (int l = 0; l < m.n_rnn; ++l) {
        const Layer& L = m.layers[l];
        MfmaLayer& M = d->L[l];
        const bool lstm = L.type == Layer::LSTM;
        const int I = L.in_size;
        M.cell = lstm ? 0 : 1;
        M.in_size = l == 0 ? I : H;                  // deeper layers contract over the padded h of the layer below
        auto weight = [&](Seg seg, int u, int g, int k) -> float { return mfma_weight(m, l, seg, u, g, k); };
        // layer 0: the 1..3 model inputs are one k-step on their own ("small" segment)
        M.w_in_off = static_cast<uint32_t>(out.size());
        if (l == 0)
            for (int w = 0; w < NW; ++w)
                for (int lane = 0; lane < kWave; ++lane)
                    for (int tl = 0; tl < TPW; ++tl) {
                        const int T = w * TPW + tl, r = lane & 15;
                        out.push_back(weight(SEG_IN, 4 * T + (r >> 2), r & 3, lane >> 4));
                    }
        // per wave one run of k-step groups: [h of the layer below (layers >= 1)] then [own h(t-1)]
        M.w_big_off = static_cast<uint32_t>(out.size());
        const int g_in = l == 0 ? 0 : H / 16, g_rec = H / 16;
        for (int w = 0; w < NW; ++w)
            for (int grp = 0; grp < g_in + g_rec; ++grp)
                for (int lane = 0; lane < kWave; ++lane)
                    for (int j = 0; j < 4; ++j)
                        for (int tl = 0; tl < TPW; ++tl) {
                            const int T = w * TPW + tl, r = lane & 15;
                            const bool in = grp < g_in;
                            const int k = 4 * (4 * (in ? grp : grp - g_in) + j) + (lane >> 4);
                            out.push_back(weight(in ? SEG_IN : SEG_REC, 4 * T + (r >> 2), r & 3, k));
                        }
        // bias of the four gate rows of every unit: the accumulators start from it
        M.b_off = static_cast<uint32_t>(out.size());
        for (int u = 0; u < H; ++u)
            for (int g = 0; g < 4; ++g) out.push_back(weight(SEG_BIAS, u, g, 0));
        M.state_off = st;
        st += static_cast<uint32_t>(lstm ? 2 * Ht : Ht);
    }
    const Layer& D = m.layers[m.n_rnn];
    d->wd_off = static_cast<uint32_t>(out.size());
    out.insert(out.end(), D.w0.begin(), D.w0.end());
    out.resize(out.size() + (H - Ht), 0.f);          // padded units carry zero Dense weights
    d->bd_off = static_cast<uint32_t>(out.size());
    out.push_back(D.w1[0]);
    // One-layer GRU: a second, GATE-MAJOR record for k_gru_gm. Above, a 16-row tile is four units x (z, r, candidate's
    // recurrent half, candidate's input half), and the fourth row meets 4-column blocks of zeros in every recurrent k-step:
    // a quarter of the matrix-core work. Here wave w owns units 16w .. 16w+15 as three recurrent tiles (z, r, recurrent
    // half: H/4 k-steps each) and the input half as a fourth that sees the input k-step only. Same weights, same
    // accumulation order per row, so the state is bit-identical to the record above. Per wave: [3][64] input k-step
    // (z, r, input half), [H/4][3][64] recurrent k-steps, [4][64][4] bias rows as the accumulators' start values.
    if (m.n_rnn == 1 && m.layers[0].type != Layer::LSTM) {
        while (out.size() % 4) out.push_back(0.f);
        d->gm_off = static_cast<uint32_t>(out.size());
        const Layer& L = m.layers[0];
        const int R = 3 * Ht, I = L.in_size;
        const float* W = L.w0.data();
        const float* U = L.w1.data();
        const float* b = L.w2.data();
        auto col = [&](int g, int u) { return (g == 0 ? 0 : g == 1 ? Ht : 2 * Ht) + u; };     // g: 0 z, 1 r, 2 recurrent half, 3 input half
        auto sc = [&](int g) { return g <= 1 ? kNegLog2e : kTwoLog2e; };                   // sigmoid rows and candidate rows, scaled as above
        auto w_in = [&](int u, int g, int k) { return (u < Ht && g != 2 && k < I) ? sc(g) * W[(size_t)k * R + col(g, u)] : 0.f; };
        auto w_rec = [&](int u, int g, int k) { return (u < Ht && g != 3 && k < Ht) ? sc(g) * U[(size_t)k * R + col(g, u)] : 0.f; };
        auto bias = [&](int u, int g) {
            if (u >= Ht) return 0.f;
            if (g <= 1) return kNegLog2e * (b[col(g, u)] + b[R + col(g, u)]);
            return kTwoLog2e * (g == 2 ? b[R + col(g, u)] : b[col(g, u)]);
        };
        for (int w = 0; w < H / 16; ++w) {
            for (int g : { 0, 1, 3 })
                for (int lane = 0; lane < kWave; ++lane) out.push_back(w_in(16 * w + (lane & 15), g, lane >> 4));
            for (int kk = 0; kk < H / 4; ++kk)
                for (int g = 0; g < 3; ++g)
                    for (int lane = 0; lane < kWave; ++lane) out.push_back(w_rec(16 * w + (lane & 15), g, 4 * kk + (lane >> 4)));
            for (int g = 0; g < 4; ++g)
                for (int lane = 0; lane < kWave; ++lane)
                    for (int e = 0; e < 4; ++e) out.push_back(bias(16 * w + 4 * (lane >> 4) + e, g));
        }
        // ... and a third record for k_gru_gs: the SAME recurrent weights (scaled as above) as A fragments of
        // v_mfma_f32_16x16x32_bf16, every fp32 weight split exactly into three bf16 terms (split_bf16x3). The fp32 matrix
        // instructions of gfx950 run at the vector rate and stall the SIMD's VALU issue while they do (profiles/r04_overlap.txt);
        // the bf16 ones are sixteen times denser, so six (or all nine) bf16 products of the split operands cost less than the
        // one fp32 product they reproduce. Per wave: [3 gates][ceil(H/32) k-steps][3 terms][64 lanes][8 bf16], lane supplies
        // row (lane & 15) = unit 16 w + (lane & 15), columns k = 32 ks + 8 (lane >> 4) + i.
        while (out.size() % 4) out.push_back(0.f);
        d->gs_off = static_cast<uint32_t>(out.size());
        const int KS2 = (H + 31) / 32;
        for (int w = 0; w < H / 16; ++w)
            for (int g = 0; g < 3; ++g)
                for (int ks = 0; ks < KS2; ++ks)
                    for (int term = 0; term < 3; ++term)
                        for (int lane = 0; lane < kWave; ++lane)
                            for (int i = 0; i < 8; i += 2) {
                                uint16_t t0[3], t1[3];
                                split_bf16x3(w_rec(16 * w + (lane & 15), g, 32 * ks + 8 * (lane >> 4) + i), t0);
                                split_bf16x3(w_rec(16 * w + (lane & 15), g, 32 * ks + 8 * (lane >> 4) + i + 1), t1);
                                const uint32_t pair = static_cast<uint32_t>(t0[term]) | (static_cast<uint32_t>(t1[term]) << 16);
                                float f;
                                std::memcpy(&f, &pair, sizeof f);
                                out.push_back(f);
                            }
    }
    // One-layer LSTM: the record of k_lstm_gs at gs_off — unit-major tiles like the GRU's gs record (wave w owns units 16w .. 16w+15
    // as four tiles, one per gate i | f | g | o; a lane's accumulator quad is one gate of four ADJACENT units, so that the new h
    // leaves as one 8-byte write per term). Per wave: [4 gates][64] input k-step (lane 16 k + r: unit 16w + r, input k), then
    // [4 gates][64 lanes][4] bias quads (lane's units 16w + 4 (lane >> 4) + e), then [4 gates][ceil(H/32) k-steps][3 terms][64 lanes]
    // [8 bf16] recurrent weights split exactly into three bf16 terms. Same scale factors as every other record (mfma_weight).
    if (m.n_rnn == 1 && m.layers[0].type == Layer::LSTM) {
        while (out.size() % 4) out.push_back(0.f);
        d->gs_off = static_cast<uint32_t>(out.size());
        const int KS2 = (H + 31) / 32;
        for (int w = 0; w < H / 16; ++w) {
            for (int g = 0; g < 4; ++g)
                for (int lane = 0; lane < kWave; ++lane) out.push_back(mfma_weight(m, 0, SEG_IN, 16 * w + (lane & 15), g, lane >> 4));
            for (int g = 0; g < 4; ++g)
                for (int lane = 0; lane < kWave; ++lane)
                    for (int e = 0; e < 4; ++e) out.push_back(mfma_weight(m, 0, SEG_BIAS, 16 * w + 4 * (lane >> 4) + e, g, 0));
            for (int g = 0; g < 4; ++g)
                for (int ks = 0; ks < KS2; ++ks)
                    for (int term = 0; term < 3; ++term)
                        for (int lane = 0; lane < kWave; ++lane)
                            for (int i = 0; i < 8; i += 2) {
                                const int u = 16 * w + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + i;
                                uint16_t t0[3], t1[3];
                                split_bf16x3(k < H ? mfma_weight(m, 0, SEG_REC, u, g, k) : 0.f, t0);
                                split_bf16x3(k + 1 < H ? mfma_weight(m, 0, SEG_REC, u, g, k + 1) : 0.f, t1);
                                const uint32_t pair = static_cast<uint32_t>(t0[term]) | (static_cast<uint32_t>(t1[term]) << 16);
                                float f;
                                std::memcpy(&f, &pair, sizeof f);
                                out.push_back(f);
                            }
        }
    }
    // Stacked models: the k-step groups of every layer once more, for k_mfma_ls — as A fragments of v_mfma_f32_16x16x32_bf16 with
    // every fp32 weight split exactly into three bf16 terms (split_bf16x3; why: see the gs record above). Same tiles (four units
    // x their four gate rows), same scale factors. Per layer: [wave][tile][segment: h of the layer below (layers >= 1) | own
    // h(t-1)][ceil(H/32) k-steps][3 terms][64 lanes][8 bf16]; lane supplies row (lane & 15), columns k = 32 ks + 8 (lane >> 4) + i.
    if (m.n_rnn >= 1) {
        while (out.size() % 4) out.push_back(0.f);
        d->ls_off = static_cast<uint32_t>(out.size());
        const int KS2 = (H + 31) / 32;
        for (int l = 0; l < m.n_rnn; ++l)
            for (int w = 0; w < NW; ++w)
                for (int tl = 0; tl < TPW; ++tl)
                    for (int seg = l == 0 ? 1 : 0; seg < 2; ++seg)
                        for (int ks = 0; ks < KS2; ++ks)
                            for (int term = 0; term < 3; ++term)
                                for (int lane = 0; lane < kWave; ++lane)
                                    for (int i = 0; i < 8; i += 2) {
                                        const int T = w * TPW + tl, r = lane & 15, k = 32 * ks + 8 * (lane >> 4) + i;
                                        uint16_t t0[3], t1[3];
                                        split_bf16x3(k < H ? mfma_weight(m, l, seg == 0 ? SEG_IN : SEG_REC, 4 * T + (r >> 2), r & 3, k) : 0.f, t0);
                                        split_bf16x3(k + 1 < H ? mfma_weight(m, l, seg == 0 ? SEG_IN : SEG_REC, 4 * T + (r >> 2), r & 3, k + 1) : 0.f, t1);
                                        const uint32_t pair = static_cast<uint32_t>(t0[term]) | (static_cast<uint32_t>(t1[term]) << 16);
                                        float f;
                                        std::memcpy(&f, &pair, sizeof f);
                                        out.push_back(f);
                                    }
    }
    *state_floats = st;
    return out;
}
bool is_conv_model(const aidax_model& m) { return m.cell == AIDAX_CELL_CONV; }

std::vector<float> pack_stack(const aidax_model& m, StackDesc* d, uint32_t* state_floats)
{
    std::vector<float> out;
    *d = StackDesc{};
    d->n_layers = m.n_rnn;
    uint32_t st = 0;
    for (int l = 0; l < m.n_rnn; ++l) {
        const Layer& L = m.layers[l];
        StackLayer& S = d->L[l];
        const int G = L.type == Layer::LSTM ? 4 : 3;
        S.cell = L.type == Layer::LSTM ? 0 : 1;
        S.in_size = L.in_size;
        S.hidden = L.out_size;
        S.rows = G * L.out_size;
        S.w_off = static_cast<uint32_t>(out.size());
        out.insert(out.end(), L.w0.begin(), L.w0.end());            // [in][rows]   (Keras kernel, already k-major)
        out.insert(out.end(), L.w1.begin(), L.w1.end());            // [hidden][rows]
        S.b_off = static_cast<uint32_t>(out.size());
        if (S.cell == 0) {
            out.insert(out.end(), L.w2.begin(), L.w2.end());
            S.b2_off = S.b_off;
        } else {
            const float* b0 = L.w2.data();
            const float* b1 = L.w2.data() + S.rows;
            for (int r = 0; r < S.rows; ++r) out.push_back(r < 2 * S.hidden ? b0[r] + b1[r] : b0[r]);
            S.b2_off = static_cast<uint32_t>(out.size());
            for (int u = 0; u < S.hidden; ++u) out.push_back(b1[2 * S.hidden + u]);
        }
        S.state_off = st;
        st += static_cast<uint32_t>(S.cell == 0 ? 2 * S.hidden : S.hidden);
        if (S.rows > d->max_rows) d->max_rows = S.rows;
        if (S.hidden > d->max_hidden) d->max_hidden = S.hidden;
        while (out.size() % 4) out.push_back(0.f);
    }
    const Layer& D = m.layers[m.n_rnn];
    d->wd_off = static_cast<uint32_t>(out.size());
    out.insert(out.end(), D.w0.begin(), D.w0.end());
    d->bd_off = static_cast<uint32_t>(out.size());
    out.push_back(D.w1[0]);
    *state_floats = st;
    return out;
}

// k_quad (aidax_quad.hip): one recurrent layer for v_mfma_f32_4x4x1_16b_f32. Wave w owns units 16w..16w+15;
// lane l supplies the weights of gate row (unit 16w + l/4, row l%4) - LSTM rows i,f,g,o; GRU rows z, r,
// recurrent half of the candidate, input half of the candidate - one register per contraction column:
//   [wave][k = 0..H-1 recurrent, H..H+2 model inputs][64 lanes], then [wave][16 units][4 rows] biases,
//   then the Dense weights [H] and bias.
std::vector<float> pack_quad(const aidax_model& m, uint32_t* bias_off, uint32_t* dense_off)
{
    const Layer& L = m.layers[0];
    const bool lstm = L.type == Layer::LSTM;
    const int H = m.hidden, I = L.in_size, G = lstm ? 4 : 3, R = G * H, NW = (H + 15) / 16;
    const float* W = L.w0.data();      // [I][R]
    const float* U = L.w1.data();      // [H][R]
    const float* b = L.w2.data();      // LSTM [R]; GRU [2][R]
    auto col_of = [&](int u, int g) { return lstm ? g * H + u : (g == 0 ? 0 : g == 1 ? H : 2 * H) + u; };
    auto rec = [&](int u, int g, int k) -> float {
        if (u >= H || (!lstm && g == 3)) return 0.f;
        return U[(size_t)k * R + col_of(u, g)];
    };
    auto inp = [&](int u, int g, int k) -> float {
        if (u >= H || k >= I || (!lstm && g == 2)) return 0.f;
        return W[(size_t)k * R + col_of(u, g)];
    };
    auto bias = [&](int u, int g) -> float {
        if (u >= H) return 0.f;
        const int c = col_of(u, g);
        if (lstm) return b[c];
        return g <= 1 ? b[c] + b[R + c] : g == 2 ? b[R + c] : b[c];
    };
    std::vector<float> out;
    for (int w = 0; w < NW; ++w)
        for (int k = 0; k < H + kMaxInputs; ++k)
            for (int lane = 0; lane < kWave; ++lane) {
                const int u = 16 * w + (lane >> 2), g = lane & 3;
                out.push_back(k < H ? rec(u, g, k) : inp(u, g, k - H));
            }
    *bias_off = static_cast<uint32_t>(out.size());
    for (int u = 0; u < 16 * NW; ++u)
        for (int g = 0; g < 4; ++g) out.push_back(bias(u, g));
    const Layer& D = m.layers[m.n_rnn];
    *dense_off = static_cast<uint32_t>(out.size());
    out.insert(out.end(), D.w0.begin(), D.w0.end());
    out.push_back(D.w1[0]);
    return out;
}

// k_lstm_q4 (aidax_q4.hip): LSTM-32 for v_mfma_f32_4x4x1_16b_f32 with the contraction split in two halves over the
// instruction's 16 blocks. Cell wave w owns units 8w..8w+7; lane l = 4 b + row: block b = (K-half b >> 3, unit 8w + (b & 7)),
// row = gate i, f, g, o (json column blocks i|f|c|o).
//   [wave][reg][64 lanes]:  reg 0       weight of the model input x (first K-half; 0 on the second)
//                           reg 1..16   recurrent weight of column 16 * half + reg - 1
//                           reg 17..20  what the accumulator rows start from on the lane as COLUMN j = l & 3 of block b:
//                                       the bias of gate (reg - 17) of the block's unit (first K-half; 0 on the second)
//   then the Dense weights [32] and bias.
std::vector<float> pack_q4(const aidax_model& m)
{
    const Layer& L = m.layers[0];
    const Layer& D = m.layers[m.n_rnn];
    const int H = 32, R = 4 * H;
    if (L.type != Layer::LSTM || m.hidden != H || L.in_size != 1 || m.n_rnn != 1) throw std::runtime_error("pack_q4: LSTM-32 with one input only");
    const float* W = L.w0.data();      // [1][R]
    const float* U = L.w1.data();      // [H][R]
    const float* b = L.w2.data();      // [R]
    std::vector<float> out(static_cast<size_t>(4) * kQ4Regs * kWave, 0.f);
    for (int w = 0; w < 4; ++w)
        for (int lane = 0; lane < kWave; ++lane) {
            const int blk = lane >> 2, row = lane & 3, half = blk >> 3, u = 8 * w + (blk & 7);
            auto reg = [&](int r) -> float& { return out[(static_cast<size_t>(w) * kQ4Regs + r) * kWave + lane]; };
            reg(0) = half == 0 ? W[row * H + u] : 0.f;
            for (int k = 0; k < 16; ++k) reg(1 + k) = U[static_cast<size_t>(16 * half + k) * R + row * H + u];
            for (int g = 0; g < 4; ++g) reg(17 + g) = half == 0 ? b[g * H + u] : 0.f;
        }
    out.insert(out.end(), D.w0.begin(), D.w0.end());
    out.push_back(D.w1[0]);
    return out;
}

// k_conv_ms (aidax_convs.hip): which tap sits at position `pos` (k-step pos / 2, channel half pos & 1) of a layer with `ksize` taps —
// taps in pairs, the oldest first; an odd count leaves the first k-step half empty ([tap 0 | padding], then [1 | 2], ...), so that the
// oldest tap, the one that may reach beyond the plane's history, has a k-step of its own. -1: padding (zero weights).
int conv_ms_tap(int ksize, int pos)
{
    if (!(ksize & 1)) return pos < ksize ? pos : -1;
    return pos == 0 ? 0 : pos == 1 ? -1 : pos - 1 < ksize ? pos - 1 : -1;
}
// Which conv stacks k_conv_ms serves: layer 0 takes the one scalar input into sixteen channels with a history of at most kConvsX0
// frames; every other layer is sixteen channels into sixteen with two to four taps; a history reaches at most kConvsFrames frames
// beyond the plane's kConvsHist (a source frame is then either in the plane or in the layer's history in HBM), and of a wave's
// four frame tiles at most two (tile, k-step) pairs read from beyond the plane (they travel in registers).
// The histories of layers 1 .. are rings over time (the frame at time tau at index tau mod hist); the stream's time is kept modulo the
// least common multiple of the history lengths, so that it never wraps out of step with any ring. 0: no common period below 2^24
// (such a stack — long, mutually prime dilations — stays on k_conv_mfma).
uint32_t conv_ms_period(const ConvDesc& d)
{
    uint64_t m = 1;
    for (int l = 1; l < d.n_layers; ++l) {
        const uint64_t h = static_cast<uint64_t>(d.L[l].hist > 0 ? d.L[l].hist : 1);
        uint64_t a = m, b = h;
        while (b) { const uint64_t t = a % b; a = b; b = t; }
        m = m / a * h;
        if (m > (1u << 24)) return 0;
    }
    return static_cast<uint32_t>(m);
}
bool conv_ms_shape_ok(const ConvDesc& d)
{
    if (d.channels != 16 || d.n_layers < 2) return false;
    if (conv_ms_period(d) == 0) return false;
    for (int l = 0; l < d.n_layers; ++l) {
        const ConvLayer& C = d.L[l];
        if (C.out_ch != 16 || C.in_ch != (l == 0 ? 1 : 16) || C.ksize < 1 || C.dilation < 1 || C.activation < 0 || C.activation > 3) return false;
        if (l == 0) { if (C.hist > kConvsX0 || C.ksize > 4) return false; continue; }
        if (C.ksize < 2 || C.ksize > 4 || C.hist > kConvsHist + kConvsFrames) return false;
        // pairs that reach beyond the plane, of wave 0 (tiles 0, 4, 8, 12: the earliest tiles reach furthest back): the kernel carries two
        // register sets for them, owned by (k-step 0, j = 0) and (k-step 0, j = 1)
        for (int ks = 0; ks < (C.ksize + 1) / 2; ++ks)
            for (int j = 0; j < 4; ++j)
                for (int h = 0; h < 2; ++h) {
                    const int tap = conv_ms_tap(C.ksize, 2 * ks + h);
                    if (tap >= 0 && 16 * (4 * j) - (C.ksize - 1 - tap) * C.dilation < -kConvsHist && (ks != 0 || j > 1)) return false;
                }
    }
    return true;
}

// Which of them k_conv_st serves as well (whole-tile fused blocks as a stream of tiles): the stacks whose geometry is compiled
// (aidax_layout.h: StGeoA ..). 1 + the geometry's index, 0 = none.
int conv_st_shape(const ConvDesc& d)
{
    if (!conv_ms_shape_ok(d)) return 0;
    for (int g = 0; g < kStGeos; ++g) {
        const bool match = st_geo_dispatch(g, [&](auto geo) {
            using G = decltype(geo);
            if (d.n_layers != G::NL) return false;
            for (int l = 0; l < G::NL; ++l)
                if (d.L[l].ksize != G::K || d.L[l].dilation != G::dil[l]) return false;
            return true;
        });
        if (match) return g + 1;
    }
    return 0;
}

std::vector<float> pack_conv(const aidax_model& m, ConvDesc* d, uint32_t* state_floats)
{
    std::vector<float> out;
    *d = ConvDesc{};
    d->n_layers = m.n_rnn;
    d->channels = m.hidden;
    uint32_t st = 0;
    for (int l = 0; l < m.n_rnn; ++l) {
        const Layer& L = m.layers[l];
        ConvLayer& C = d->L[l];
        C.in_ch = L.in_size; C.out_ch = L.out_size; C.ksize = L.ksize; C.dilation = L.dilation;
        C.activation = L.activation;
        C.hist = (L.ksize - 1) * L.dilation;
        C.w_off = static_cast<uint32_t>(out.size());
        out.insert(out.end(), L.w0.begin(), L.w0.end());
        C.b_off = static_cast<uint32_t>(out.size());
        out.insert(out.end(), L.w1.begin(), L.w1.end());
        C.state_off = st;
        st += (static_cast<uint32_t>(C.hist * C.in_ch) + 3u) & ~3u;     // every layer's history starts 16-byte aligned (float4 copies)
        if (C.hist > d->max_hist) d->max_hist = C.hist;
        // matrix-core form: the layer as a [frames x (tap,cin)] . [(tap,cin) x cout] contraction
        const int K = C.ksize * C.in_ch;
        C.k_steps = (K + 3) / 4;
        if (C.k_steps > d->max_k_steps) d->max_k_steps = C.k_steps;
        // one 8-byte record per (k-step, lane): the B fragment value K[tap][cin][cout = lane & 15] of contraction row
        // k = 4*kk + (lane >> 4) = tap*in_ch + cin (zero past ksize*in_ch / out_ch), and where that row lives in the
        // activation plane as (cin << 16 | frames back); padding rows (zero weights) point at row 0 so they read valid data
        while (out.size() % 4) out.push_back(0.f);
        C.wf_off = static_cast<uint32_t>(out.size());
        // (tanh layers: weights and bias times 2 log2 e, see ConvLayer)
        const float scale = C.activation == 1 ? kTwoLog2e : 1.0f;
        for (int kk = 0; kk < C.k_steps; ++kk)
            for (int lane = 0; lane < kWave; ++lane) {
                const int k = 4 * kk + (lane >> 4), co = lane & 15;
                out.push_back((k < K && co < C.out_ch) ? scale * L.w0[(size_t)k * C.out_ch + co] : 0.f);   // w0 is [tap][cin][cout]
                const int kc = k < K ? k : 0;
                const int32_t where = ((kc % C.in_ch) << 16) | ((C.ksize - 1 - kc / C.in_ch) * C.dilation);
                float f;
                std::memcpy(&f, &where, 4);
                out.push_back(f);
            }
        C.bs_off = static_cast<uint32_t>(out.size());
        for (int co = 0; co < 16; ++co) out.push_back(co < C.out_ch ? scale * L.w1[co] : 0.f);
    }
    // the full-block records: the plane geometry needs the longest history of the stack, known only now
    {
        const int F = convm_plane_stride(d->max_hist, kConvmFullFrames), Hb = (d->max_hist + 3) & ~3;
        for (int l = 0; l < m.n_rnn; ++l) {
            ConvLayer& C = d->L[l];
            while (out.size() % 4) out.push_back(0.f);
            const uint32_t from = C.wf_off;
            C.wf_full_off = static_cast<uint32_t>(out.size());
            for (int i = 0; i < C.k_steps * kWave; ++i) {
                const float b = out[from + 2 * i];
                int32_t where;
                std::memcpy(&where, &out[from + 2 * i + 1], 4);
                const int32_t byte_off = 4 * ((where >> 16) * F + convm_swz(where >> 16, Hb - (where & 0xffff) + (i & 15)));
                float f;
                std::memcpy(&f, &byte_off, 4);
                out.push_back(b);
                out.push_back(f);
            }
        }
    }
    // k_conv_ms's records (ConvLayer::ms_*), for the stacks it serves: one scalar input into sixteen channels, then sixteen into
    // sixteen; up to four taps (two bf16 k-steps); histories the kernel's plane + at most two deep fragment pairs per wave can hold.
    d->ms_ok = conv_ms_shape_ok(*d) ? 1 : 0;
    if (d->ms_ok) {
        uint32_t mst = 0;
        for (int l = 0; l < m.n_rnn; ++l) {
            const Layer& L = m.layers[l];
            ConvLayer& C = d->L[l];
            C.ms_state_off = mst;
            mst += l == 0 ? ((static_cast<uint32_t>(C.hist) + 3u) & ~3u) : static_cast<uint32_t>(C.hist) * 24u;      // 96 bytes per frame of history
            C.ms_ksteps = 0; C.ms_w_off = 0;
            for (auto& r : C.ms_shift) for (auto& v : r) v = -1;
            if (l == 0) continue;                                   // layer 0 is a few FMAs per output: the fp32 records above serve it
            // taps in pairs, the oldest first; an odd count leaves the FIRST k-step half empty (the oldest tap — the one that may
            // reach beyond the plane's history — then has a k-step of its own)
            int taps[2][2] = { { -1, -1 }, { -1, -1 } };
            const int odd = C.ksize & 1;
            C.ms_ksteps = (C.ksize + 1) / 2;
            for (int pos = 0; pos < 2 * C.ms_ksteps; ++pos) taps[pos / 2][pos & 1] = conv_ms_tap(C.ksize, pos);
            // An odd count's lone first tap does not leave half of its k-step to padding: its SIX term products go out as THREE matrix
            // instructions, two products side by side in the two halves of the k-step — [w0 | w1] x [x0 | x0], [w2 | w0] x [x0 | x1],
            // [w1 | w0] x [x1 | x2] (ms_packed0: both halves read the tap's frame, a lane's B term depends on its half). Nine matrix
            // instructions per tile and three-tap layer instead of twelve.
            C.ms_packed0 = odd ? 1 : 0;
            if (odd) taps[0][1] = taps[0][0];
            for (int ks = 0; ks < C.ms_ksteps; ++ks)
                for (int h = 0; h < 2; ++h) C.ms_shift[ks][h] = taps[ks][h] < 0 ? int16_t(-1) : static_cast<int16_t>((C.ksize - 1 - taps[ks][h]) * C.dilation);
            const float scale = C.activation == 1 ? kTwoLog2e : 1.0f;
            while (out.size() % 4) out.push_back(0.f);
            C.ms_w_off = static_cast<uint32_t>(out.size());
            for (int ks = 0; ks < C.ms_ksteps; ++ks)
                for (int term = 0; term < 3; ++term)
                    for (int lane = 0; lane < kWave; ++lane) {
                        const int co = lane & 15, q = lane >> 4, tap = taps[ks][q >> 1];
                        for (int i = 0; i < 8; i += 2) {
                            uint16_t t0[3] = { 0, 0, 0 }, t1[3] = { 0, 0, 0 };
                            if (tap >= 0) {
                                const int ci = 8 * (q & 1) + i;
                                split_bf16x3(scale * L.w0[((size_t)tap * C.in_ch + ci) * C.out_ch + co], t0);
                                split_bf16x3(scale * L.w0[((size_t)tap * C.in_ch + ci + 1) * C.out_ch + co], t1);
                            }
                            // (the packed first k-step: record `term` is instruction `term` — weight term kPackedW[term][half of the k-step])
                            static const int kPackedW[3][2] = { { 0, 1 }, { 2, 0 }, { 1, 0 } };
                            const int wt = (ks == 0 && odd) ? kPackedW[term][q >> 1] : term;
                            const uint32_t pair = static_cast<uint32_t>(t0[wt]) | (static_cast<uint32_t>(t1[wt]) << 16);
                            float f;
                            std::memcpy(&f, &pair, sizeof f);
                            out.push_back(f);
                        }
                    }
        }
        d->ms_pos_off = mst;                                        // the stream's time (uint32), one 16-byte slot
        mst += 4u;
        d->ms_pos_mod = conv_ms_period(*d);
        d->st_ok = conv_st_shape(*d);
#ifdef AIDAX_CONV_TRACE
        if (d->st_ok) { d->st_trace_off = mst; mst += 256u; }       // the measurement build's stamps (k_conv_st)
#endif
        d->ms_state_floats = mst;
    }
    const Layer& D = m.layers[m.n_rnn];
    d->wd_off = static_cast<uint32_t>(out.size());
    out.insert(out.end(), D.w0.begin(), D.w0.end());
    d->bd_off = static_cast<uint32_t>(out.size());
    out.push_back(D.w1[0]);
    *state_floats = st;
    return out;
}

std::vector<float> pack_weights(const aidax_model& m)
{
    if (m.n_rnn == 1 && m.cell == AIDAX_CELL_LSTM) return pack_lstm(m);
    if (m.n_rnn == 1 && m.cell == AIDAX_CELL_GRU) return pack_gru(m);
    throw std::runtime_error("no packer for this architecture");
}

}  // namespace aidax
