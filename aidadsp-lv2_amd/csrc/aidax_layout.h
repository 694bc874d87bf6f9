// aidax_layout.h — data layout shared by the host packer and the HIP kernels.
//
// HBM layout (all per pool, device-resident across process() calls):
//   StreamCtl   ctl[n_streams]     control-rate values latched from the ports (host-written)
//   StreamState st[n_streams]      biquad z1/z2, smoother memories, pending one-shots
//   float       nn[n_streams][NN_STATE]   recurrent state of the model (h, c per layer)
//   float       wpack[...]         weights, pre-shuffled into the register order of the
//                                  kernel's lane mapping: element (idx, lane) at idx*64+lane,
//                                  so a wave loads each weight register with one coalesced
//                                  256-byte read.
//   audio in/out [n_streams][n_frames] fp32, one row per stream (the C-ABI layout).
#pragma once

#include <cstdint>

// Test and measurement switches (AIDAX_TUNE, AIDAX_KERNEL, AIDAX_LP_COOP, ... and the kernels' LaunchArgs::tune bits) exist only in a
// library built with -DAIDAX_TEST_HOOKS: the build the test suite and scratch/ load (make all -> lib/hooks/libaidax_hip.so). The shipped
// library (lib/libaidax_hip.so: bench.py, smoke(), the LV2 shell, the bundle) compiles AIDAX_HOOK_ENV to a null pointer and AIDAX_TUNE
// to 0: none of the names is in its strings, none of the switched code in its kernels. What the shipped library does read from the
// environment is configuration, documented in INTEGRATION.md: AIDAX_ZEROCOPY, AIDAX_SPIN_WAIT (pool), AIDAX_DEVICE, AIDAX_HUB,
// AIDAX_HUB_FRAMES, AIDAX_HUB_DEADLINE_US, AIDAX_STRICT_REFERENCE_SET (LV2 shell / loader).
#ifdef AIDAX_TEST_HOOKS
#define AIDAX_HOOK_ENV(name) std::getenv(name)
#define AIDAX_TUNE(a) ((a).tune)
#else
#define AIDAX_HOOK_ENV(name) (static_cast<const char*>(nullptr))
#define AIDAX_TUNE(a) 0
#endif

namespace aidax {

constexpr int kWave = 64;
constexpr int kMaxInputs = 3;            // MAX_INPUT_SIZE, model_variant.hpp:4

// Biquad slots of one stream, in chain order of use.
enum BqSlot { BQ_LPF = 0, BQ_DC = 1, BQ_DEPTH = 2, BQ_BASS = 3, BQ_MID = 4, BQ_TREBLE = 5, BQ_PRESENCE = 6, BQ_COUNT = 7 };

// StreamCtl.flags
enum : uint32_t {
    CTL_ENABLED   = 1u << 0,   // *enabled > 0.5                      (rt-neural-generic.cpp:495)
    CTL_LPF_ON    = 1u << 1,   // in_lpf_pc != 0                      (:622)
    CTL_EQ_PRE    = 1u << 2,   // eq_position == 1 && eq_bypass == 0  (:628)
    CTL_EQ_POST   = 1u << 3,   // eq_position == 0 && eq_bypass == 0  (:651)
    CTL_EQ_BANDPASS = 1u << 4, // mid_type == BANDPASS: mid only      (:130-132)
    CTL_NET_ON    = 1u << 5,   // model && !net_bypass                (:631-632)
    CTL_DC_ON     = 1u << 6,   // *dc_blocker_param == 1              (:646)
};

// StreamState.pending (one-shots, consumed by the next process pass)
enum : uint32_t {
    PEND_ACTIVATE   = 1u << 0,   // activate(): gain smoothers clearToTargetValue (:341-342)
    PEND_PARAM_FIRST = 1u << 1,  // DynamicModel::paramFirstRun (:350, :636-640, :1061)
};

struct alignas(16) StreamCtl {
    double   bq[BQ_COUNT][5];     // a0 a1 a2 b1 b2 (common/Biquad.h:48)
    float    pre_target;          // DB_CO(pregain_db)                    (:489)
    float    master_target;       // loading ? 0 : DB_CO(master_db)       (:490, :654)
    float    pre_coef;            // ExponentialValueSmoother::coef, host rate (:283-290)
    float    master_coef;
    float    p_target[2];         // PARAM1 / PARAM2                      (:634-635)
    float    p_den;               // tau * model samplerate               (ValueSmoother.hpp:239)
    uint32_t flags;
    uint32_t pad[2];
};
static_assert(sizeof(StreamCtl) == 320, "StreamCtl layout");

struct alignas(16) StreamState {
    double   z[BQ_COUNT][2];      // z1 z2 (common/Biquad.h:50)
    float    pre_mem, master_mem; // ExponentialValueSmoother::mem
    float    pre_tgt, master_tgt; // ExponentialValueSmoother::target as last set
    float    p_mem[2], p_step[2], p_tgt[2];   // LinearValueSmoother mem/step/target
    uint32_t pending;
    uint32_t pad;
};
static_assert(sizeof(StreamState) == 160, "StreamState layout");

// ---------------------------------------------------------------- lane mapping
// One wavefront (64 lanes) per stream. A recurrent layer with G gates and H
// units is spread as S lanes per unit slot: lane = part*(64/S) + slot,
// unit j = slot + m*(64/S) for m in [0,NU), each lane holding GPL = G/S gate
// rows of each of its units (gate = part + S*e, e in [0,GPL)). Every row keeps
// its H recurrent weights, 3 input weights and bias in registers.
struct LaneMap {
    int S, slots, NU, GPL;
    constexpr int rows_per_lane() const { return NU * GPL; }
    constexpr int regs_per_row(int H) const { return H + kMaxInputs + 1; }
};

constexpr int ceil_div(int a, int b) { return (a + b - 1) / b; }

constexpr LaneMap lstm_map(int H)
{
    // pick S in {4,2,1} minimising FMA issue NU*GPL*(H+3); ties -> fewer lanes per unit
    int best_S = 1, best_cost = 1 << 30;
    for (int S = 1; S <= 4; S *= 2) {
        const int slots = kWave / S;
        const int NU = ceil_div(H, slots);
        const int cost = NU * (4 / S) * (H + kMaxInputs) + NU * 24 /* activations weigh in */;
        if (cost < best_cost) { best_cost = cost; best_S = S; }
    }
    return LaneMap{best_S, kWave / best_S, ceil_div(H, kWave / best_S), 4 / best_S};
}

constexpr LaneMap gru_map(int H)
{
    // GRU: the S lanes of a unit each take H/S columns of all three rows (K-split) and add their
    // partial sums with permlane swaps. Estimated issue slots per sample: NU*(3*H/S + 9*log2(S) + 25);
    // S = 2 wins for H in {12..32} and for H = 80 (where S = 1 would need 504 weight registers).
    int best_S = 1, best_cost = 1 << 30;
    for (int S = 1; S <= 2; S *= 2) {
        if ((H / S) % 4 != 0) continue;
        const int NU = ceil_div(H, kWave / S);
        const int cost = NU * (3 * (H / S) + (S == 2 ? 9 : 0) + 25);
        if (cost < best_cost) { best_cost = cost; best_S = S; }
    }
    return LaneMap{best_S, kWave / best_S, ceil_div(H, kWave / best_S), 3};
}

// Packed weight record, index -> meaning, for one recurrent layer:
//   for m in [0,NU) for e in [0,GPL):  H recurrent weights (k = 0..H-1), 3 input weights, 1 bias
//   GRU only: + for m: bias_n1 (recurrent bias of the candidate row, kept apart)
//   then for m in [0,NU): dense weight of unit (0 outside part 0 / j >= H)
//   then 1: dense bias
// Cells whose units fill whole 16-lane rows read h(t-1) out of the neighbours' registers (row rotations, LstmCell::ROT /
// ROT4) in the latency-bound forms — one wave per stream, the pipeline: LSTM-32 (S = 2: two rows of units, half of the
// recurrent FMAs) and LSTM-8 / 12 / 16 (S = 4: one row of units per gate, all of them; a row then carries 16 weights, zeros
// for units beyond H). Their record comes twice: [lane records, recurrent weights in rotation order | Dense tail | lane
// records in natural order] — the throughput-bound k_nn, where the dearer DPP FMAs cost more than the LDS latency they hide,
// reads the second copy with broadcast LDS reads as before.
constexpr bool lstm_rot4(int H) { return lstm_map(H).S == 4 && lstm_map(H).NU == 1; }
constexpr bool lstm_rot2(int H) { return lstm_map(H).S == 2 && H == 32; }
constexpr bool lstm_has_alt_pack(int H) { return lstm_rot2(H) || lstm_rot4(H); }
constexpr int lstm_row_weights(int H, bool rotation_order) { return (rotation_order && lstm_rot4(H)) ? 16 : H; }
constexpr int lstm_pack_regs(int H, bool rotation_order)
{
    const LaneMap L = lstm_map(H);
    return L.NU * L.GPL * (lstm_row_weights(H, rotation_order) + kMaxInputs + 1) + L.NU + 1;
}
constexpr int lstm_pack_regs(int H) { return lstm_pack_regs(H, lstm_has_alt_pack(H)); }       // the first (primary) record
constexpr int lstm_alt_pack_offset(int H) { return lstm_pack_regs(H) * kWave + H + 1; }      // floats from the record's start

constexpr int gru_pack_regs(int H)
{
    const LaneMap L = gru_map(H);
    return L.NU * 3 * (H / L.S + kMaxInputs + 1) + L.NU + L.NU + 1;
}

// Kernel run modes
enum : int {
    MODE_CHAIN = 0,     // full run() chain on audio blocks
    MODE_WARMUP = 1,    // applyModel over zeros, output discarded (:1077-1078)
    MODE_NN_ONLY = 2,   // bare applyModel with explicit [n][I] inputs (self-test / forward)
};

// ---- descriptors of the extension kernels (aidax_stack.hip); offsets index the weight buffer
constexpr int kStackStreams = 8;      // streams sharing one workgroup (and every weight load) in k_stack
constexpr int kMaxStackLayers = 4;
constexpr int kMaxConvLayers = 12;

struct StackLayer {
    int32_t cell;          // 0 LSTM, 1 GRU
    int32_t in_size, hidden, rows;
    uint32_t w_off;        // Wt[in_size + hidden][rows]: input rows first, then recurrent rows (Keras layout as is)
    uint32_t b_off;        // [rows]   LSTM: b; GRU: z,r -> b0+b1, candidate -> b0
    uint32_t b2_off;       // GRU: recurrent-side candidate bias [hidden]
    uint32_t state_off;    // into the stream's nn state: h[hidden] (+ c[hidden] for LSTM)
};
struct StackDesc {
    int32_t n_layers, max_rows, max_hidden, pad;
    StackLayer L[kMaxStackLayers];
    uint32_t wd_off, bd_off;
};

// k_mfma (aidax_mfma.hip): the same models on the matrix cores. One workgroup of four waves carries
// kMfmaStreams streams as the N dimension of v_mfma_f32_16x16x4_f32; the weights are stored as
// ready-made A fragments (one coalesced load per wave instruction):
//   tile T = 16 gate rows = 4 units x 4 rows (LSTM i,f,g,o; GRU z, r, n_recurrent, n_input), wave w
//   owns tiles [w*TPW, (w+1)*TPW) with TPW = hidden/4/waves (waves = 8 when hidden % 32 == 0, else 4); lane supplies row (lane&15), k = 4*kk + (lane>>4).
//   k-steps come in groups of 4:  [wave][group][lane][4 k-steps][TPW]   -> TPW float4 per lane and group;
//   a wave's groups run [h of the layer below | own h(t-1)]; layer 0's 1..3 inputs are one k-step on
//   their own, [wave][lane][TPW]; the bias is a plain [unit][4 rows] table the accumulators start from.
// k_lstm_q4 (aidax_q4.hip): registers per lane and cell wave of its weight record — 17 k-step weights, 4 accumulator
// start values (the bias rows, on the lanes of the first K-half)
constexpr int kQ4Regs = 21;

constexpr int kMfmaStreams = 16;
#ifndef AIDAX_MFMA_WIDE
#define AIDAX_MFMA_WIDE 1
#endif
// waves per k_mfma workgroup: 8 (two per SIMD) when the tiles divide, else 4
constexpr int mfma_waves(int hidden) { return (AIDAX_MFMA_WIDE && hidden % 32 == 0) ? 8 : 4; }
struct MfmaLayer {
    int32_t cell;          // 0 LSTM, 1 GRU
    int32_t in_size;       // layer 0: 1..3 model inputs (one small segment); deeper: hidden of the layer below
    uint32_t w_in_off;     // layer 0 only: the model-input k-step
    uint32_t w_big_off;    // the k-step groups
    uint32_t b_off;        // [hidden][4]
    uint32_t state_off;    // as StackLayer
};
struct MfmaDesc {
    int32_t n_layers, hidden, tpw, waves;      // hidden = the kernel's width: the model's rounded up to 16 (zero rows / columns)
    int32_t hidden_true;                       // the model's width: layout of the recurrent state in HBM (h[.] then c[.])
    uint32_t gm_off;                           // one-layer GRU: offset of the gate-major record k_gru_gm reads (0: none), see pack_mfma
    uint32_t gs_off;                           // ... and of its bf16 x 3 twin k_gru_gs reads (recurrent weights as split bf16 fragments)
    uint32_t ls_off;                           // stacked models: offset of the split record k_mfma_ls reads (every layer's k-step groups as bf16 x 3 fragments; 0: none)
    MfmaLayer L[kMaxStackLayers];
    uint32_t wd_off, bd_off;
};

// k_quad (aidax_quad.hip): offsets into the weight buffer written by pack_quad
struct QuadDesc {
    uint32_t bias_off;     // [16 * waves][4] gate-row biases
    uint32_t dense_off;    // Dense weights [hidden] + bias
};

struct ConvLayer {
    int32_t in_ch, out_ch, ksize, dilation, activation;
    int32_t hist;          // (ksize-1)*dilation frames of input history
    uint32_t w_off;        // K[ksize][in_ch][out_ch]
    uint32_t b_off;        // [out_ch]
    uint32_t state_off;    // history[in_ch][hist] in the stream's nn state
    // k_conv_mfma: [k_steps][64 lanes] records {B fragment value of v_mfma_f32_16x16x4_f32, (cin << 16 | frames back)}:
    // lane supplies K[tap][cin][cout = lane&15] for k = 4*kk + (lane>>4) = tap*in_ch + cin (zero past ksize*in_ch / out_ch)
    uint32_t wf_off;
    int32_t k_steps;
    // ... the same records for a block of exactly kConvmFullFrames frames with the second word FINAL — the byte offset of the A
    // value in the activation plane, which depends on the plane geometry and so on the block length — so that the kernel's
    // full-block instantiation stages them as they are; and the bias the matrix-core kernel starts its accumulators from.
    // Both carry the factor 2 log2(e) on tanh layers: tanh(x) = 1 - 2 / (1 + 2^(x 2 log2 e)) then costs four instructions.
    uint32_t wf_full_off;
    uint32_t bs_off;       // [16]
    // k_conv_ms (aidax_convs.hip): the layer as bf16 term products on v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand —
    // [ms_ksteps][3 terms][64 lanes] fragments of 8 bf16: lane supplies row (lane & 15) = cout, columns 8 (lane >> 4) .. + 7 of a
    // k-step of 32 = two taps x sixteen input channels (quarter q: tap ms_tap[ks][q >> 1], channels 8 (q & 1) ..); every weight
    // (times 2 log2 e on tanh layers, like the records above) split exactly into three bf16 terms. ms_shift[ks][h]: frames back
    // of the tap in half h of k-step ks, -1 = padding (zero weights). ms_state_off: this layer's input history in the stream's nn
    // state in the layout of the kernel's activation plane ([3 terms][2 channel halves][hist frames][8 bf16]; layer 0: fp32 [hist]) — every
    // strip a RING over time: the frame at time tau of the stream at index tau mod hist (ConvDesc::ms_pos_off holds the time).
    uint32_t ms_w_off;
    int32_t  ms_ksteps;
    int16_t  ms_shift[2][2];
    int32_t  ms_packed0;   // three taps: k-step 0 is the oldest tap's six term products in three instructions (see pack_conv); B terms per half: {0,0,1} | {0,1,2}
    uint32_t ms_state_off;
};
// k_conv_ms's activation plane: per (term, channel half) a strip of kConvsPF frames x 16 bytes — kConvsHist frames of history in
// front of the block's 256. A layer whose history is longer reads the part beyond it straight from HBM (as B fragments).
constexpr int kConvsHist = 128, kConvsFrames = 256, kConvsPF = kConvsHist + kConvsFrames;
constexpr int kConvsX0 = 16;             // frames of layer 0's (scalar, fp32) input history in front of the audio row
// k_conv_mfma's activation plane: frames per channel row = history (rounded to 4) + block (rounded to whole tiles), then up to
// 16 (mod 64) floats (the four cin groups of an A fragment on disjoint banks)
constexpr int kConvmFullFrames = 256;
constexpr int convm_plane_stride(int max_hist, int n_frames)
{
    return ((((max_hist + 3) & ~3) + ((n_frames + 15) & ~15) + 47) / 64) * 64 + 16;
}
// ... and its columns are SWIZZLED per channel: inside every aligned group of sixteen columns the four 16-byte blocks trade places
// by (channel >> 2) & 3. Why: the tanh outputs leave as one ds_write_b128 per lane — four consecutive frames of one channel, the
// sixteen lanes of a quarter-wave holding channels 0 .. 15 — and with the row stride = 16 (mod 64) floats that the A-fragment
// reads need, channels c, c + 4, c + 8, c + 12 start in the same bank: a four-way conflict on every output write (31 % of the
// kernel's LDS cycles, profiles/r04_cfg4_pmc_summary.txt). With the swizzle those four land in four different 16-byte blocks
// of the bank window. The reads stay conflict-free: sixteen consecutive columns still cover sixteen different banks (the
// swizzle permutes blocks inside an aligned group, and two groups are two bank windows), and the four channels of an A
// fragment (4 kk + q) share their (channel >> 2). Adding a multiple of 16 columns commutes with it (the tile offsets stay
// immediates).
constexpr int convm_swz(int ch, int col) { return col ^ (((ch >> 2) & 3) << 2); }
constexpr float kTwoLog2e = 2.88539008177792681472f;
struct ConvDesc {
    int32_t n_layers, channels, max_hist, max_k_steps;
    ConvLayer L[kMaxConvLayers];
    uint32_t wd_off, bd_off;
    int32_t  ms_ok;             // k_conv_ms serves this stack (pack_conv: conv_ms_shape_ok)
    uint32_t ms_state_floats;   // ... with this much state per stream (its history layout differs from k_conv's / k_conv_mfma's)
    uint32_t ms_pos_off;        // ... in it one uint32: the time of the next block's first frame, modulo ms_pos_mod — where the histories' rings stand
    uint32_t ms_pos_mod;        // the least common multiple of the layers' history lengths (conv_ms_shape_ok bounds it)
    int32_t  st_ok;             // k_conv_st (the streaming form of whole-tile fused blocks) serves this stack: 1 + the index of its geometry (conv_st_shape)
    uint32_t st_trace_off;      // (measurement build: k_conv_st's stamps behind the histories)
};

// k_conv_st's GEOMETRIES (aidax_convs.hip): the stacks whose streaming form is compiled — layer count, taps per layer, the layers' dilations
// (powers of two: a history length is then one too, and a ring index a mask) and which of the three layer waves runs which layers
// (wave w = 1 .. 3: layers [wbeg[w - 1], wbeg[w]); layer 0, the scalar input layer, rides on wave 1). Everything else — which taps live in
// the LDS ring in front of a layer and which come from HBM, ring lengths and offsets, the staging slots — follows from these at compile time
// (StG<G> in aidax_convs.hip); conv_st_shape() (aidax_pack.cpp) matches a model against this list.
struct StGeoA { static constexpr int NL = 8,  K = 3; static constexpr int dil[kMaxConvLayers] = { 1, 2, 4, 8, 16, 32, 64, 128 };      static constexpr int wbeg[4] = { 1, 3, 6, 8 }; };   // BASELINE cfg4
struct StGeoB { static constexpr int NL = 10, K = 2; static constexpr int dil[kMaxConvLayers] = { 1, 2, 4, 8, 16, 1, 2, 4, 8, 16 };    static constexpr int wbeg[4] = { 1, 4, 7, 10 }; };  // two cycles 1 .. 16, two taps
struct StGeoC { static constexpr int NL = 6,  K = 3; static constexpr int dil[kMaxConvLayers] = { 1, 2, 4, 8, 16, 32 };                static constexpr int wbeg[4] = { 1, 2, 4, 6 }; };   // a shallow stack
struct StGeoD { static constexpr int NL = 8,  K = 3; static constexpr int dil[kMaxConvLayers] = { 1, 2, 4, 8, 1, 2, 4, 8 };            static constexpr int wbeg[4] = { 1, 3, 6, 8 }; };   // two cycles 1 .. 8, three taps
struct StGeoE { static constexpr int NL = 8,  K = 2; static constexpr int dil[kMaxConvLayers] = { 1, 2, 4, 8, 16, 32, 64, 128 };      static constexpr int wbeg[4] = { 1, 3, 6, 8 }; };   // cfg4's dilations, two taps
constexpr int kStGeos = 5;
template <class F> constexpr auto st_geo_dispatch(int g, F&& f)     // f(G{}) for geometry index g
{
    switch (g) {
    case 0: return f(StGeoA{});
    case 1: return f(StGeoB{});
    case 2: return f(StGeoC{});
    case 3: return f(StGeoD{});
    default: return f(StGeoE{});
    }
}

struct LaunchArgs {
    const StreamCtl* ctl;
    StreamState*     st;
    float*           nn;          // [n_streams][nn_stride]
    const float*     wpack;
    const float*     in;          // MODE_CHAIN: [n_streams][n]; MODE_NN_ONLY: [n][I] (stream 0)
    float*           out;
    uint32_t         n_streams;
    uint32_t         n_frames;
    uint32_t         nn_stride;
    int32_t          mode;
    int32_t          input_size;
    int32_t          input_skip;
    float            in_gain, out_gain;
    int32_t          tune;        // AIDAX_TUNE bit mask, measurement / test switches of the kernels (0 in production):
                                  // 1 = no issue priority for the recurrent wave of k_*_pipe, 2 = k_mfma_lp with a group's layers on adjacent workgroup ids,
                                  // 16 = k_mfma_lp reports a hand-over give-up that did not happen (tests of the fault path),
                                  // 32768 / 65536 = k_conv_st without its history stores / with every far tap reading one resident tile (scratch/r06_cfg4_ablate.sh),
                                  // 8192 = the last layer's workgroup of k_mfma_lp's one-launch form starts 100 us late (tests: nothing may lean on the layers' workgroups starting together)
    uint32_t         row_stride;  // k_conv_mfma: frames between two streams' rows in `in` / `out` when a launch carries a
                                  // time slice of a longer block (0: rows are n_frames apart)
    uint32_t         done_seq;    // k_*_pipe / k_*_pipe4 of a ONE-stream pool whose `out` is the host's memory: the wave that has stored the block
    uint32_t*        done_word;   // writes done_seq here behind it (host memory; nullptr: nobody waits this way) — the blocking path's completion
                                  // word without a packet behind the pass (aidax_pool_process, profiles/r06_host_pipeline.txt)
};

}  // namespace aidax
