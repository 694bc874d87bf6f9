// aidax_pool.cpp — the stream pool behind the C ABI: device residency, control
// latching, model swap and the process launches. Host logic only; the kernels
// are in aidax_kernels.hip. No CPU fallback: any HIP failure is AIDAX_ERR_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <sstream>

#include "aidax_internal.h"
#include "aidax_kernels.h"

using namespace aidax;

namespace aidax {

// Which models have a device path: the reference's 54 variants (18 register-resident kernels; k_quad /
// k_mfma for many streams), stacked or wider recurrent layers (k_mfma, k_stack) and conv1d stacks
// (k_conv_mfma, k_conv) as extensions.
bool model_supported(const aidax_model& m)
{
    if (is_conv_model(m)) {
        if (m.input_size != 1 || m.hidden > 16 || m.n_rnn > kMaxConvLayers) return false;
        for (int l = 0; l < m.n_rnn; ++l)
            if (m.layers[l].out_size != m.hidden || m.layers[l].ksize > 8) return false;
        return true;
    }
    if (is_stack_model(m)) return m.n_rnn <= kMaxStackLayers && m.hidden <= 128 && m.hidden % 4 == 0;
    return m.n_rnn == 1 && find_kernel(m.cell, m.hidden) != nullptr;
}

namespace {

constexpr uint32_t kWarmupFrames = 2048;        // rt-neural-generic.cpp:1077
constexpr uint32_t kMaxFrames = 8192;           // LDS block buffer bound (32 KiB)
// Which many-streams form serves a model of the reference's table best at a given stream count. Measured
// (scratch/perf_table_mfma.py, 256-frame blocks, MI355X): k_quad (4 streams per workgroup on
// mfma_4x4x1, weights in registers) wins for <= 32 units from 4096 streams (1.15-1.5x over the split form) and
// for the wide cells at every stream count below that (LSTM-64 1.3-1.6x, LSTM-80 1.3-1.75x, GRU-80 1.2-1.4x);
// k_mfma (16 streams per workgroup on mfma_16x16x4) takes over for the wide cells from 4096-8192 streams
// (LSTM-80 @ 16384: 2.3 ms against 4.3 ms for k_quad and 8.4 ms for the one-wave kernel). GRU-64 ties with
// its register kernel up to 8192 streams and stays there.
// Round 3 (scratch/perf_lp1.py, profiles/r03_cfg3_forms.txt): the one-launch form of k_mfma_lp (one workgroup per 16
// streams, the DSP chain on its helper waves) costs the same whatever the stream count while its workgroups fit the
// machine in one round, and beats every other form when that round is (nearly) full — LSTM-32 209 us against 231,
// GRU-32 197 / 211, LSTM-40 338 / 346, GRU-64 434 / 468 at 4096 streams on 256 CUs; at 3072 streams the others win.
enum ManyForm { MANY_NONE = 0, MANY_QUAD = 1, MANY_MFMA = 2 };
// One-layer models on k_mfma_ls1 (k_mfma_ls's body for a lone layer: the recurrent product as bf16 term products, a workgroup per
// 16 streams): where it beats every other form at 256-frame blocks on 256 CUs (scratch/ls1_ab.py, profiles/r04_ls1_ab.txt) —
// LSTM-64 from 2048 streams (4096: 306 us against 417 on k_mfma_lp, 16 384: 1 228 against 1 609 on k_mfma), LSTM-80 / GRU-80 from
// 1024 (4096: 540 / 458 against 716 / 669), LSTM-40 beyond 4096 (8192: 427 against 526), 32 units beyond 6144 (8192: 286 / 248
// against 345 / 289 on k_quad). GRU-40 / 64 have k_gru_gs (190 us per round of 4096 streams against 244 .. 275 here); 16 units never.
// (A GRU-80 pool this rule sends to the matrix-core forms runs k_gru_gs<5, 2> — 324 us per 4096 streams against 436 here — and LSTM-40 / 64
// have k_lstm_gs, below; k_mfma_ls1 then serves LSTM-80, 32 units at many streams, and the A/B runs.)
// Round 5 (profiles/r05_blocklen_forms*.txt: the table re-measured at 64- and 128-frame blocks, what a host's period really is): the
// crossovers hold at every block length but two, both wrong at 256 frames too — see many_streams_form — and ONE moves with the block:
// LSTM-32 at 5632 .. 6144 streams is k_mfma_ls1's from 192-frame blocks (268 against 278 us on k_nn), k_nn's below (84 / 149 us at 64 /
// 128 frames against 93 / 152). `max_frames` is the pool's block length.
bool lone_split_pays(int cell, int hidden, uint32_t n, int cus, uint32_t max_frames)
{
    const bool off = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_LP_SPLIT"); return e && e[0] == '0'; }();
    if (off || cus <= 0) return false;
    const bool lstm = cell == AIDAX_CELL_LSTM;
    const uint32_t groups = (n + kMfmaStreams - 1) / kMfmaStreams, c = static_cast<uint32_t>(cus);
    switch (hidden) {
    case 32: return groups * 2 > c * 3 || (lstm && max_frames >= 192 && groups * 8 >= c * 11);
    case 40: return lstm && groups > c;
    case 64: return lstm && groups * 2 > c;
    case 80: return groups * 4 > c;
    default: return false;
    }
}

// ... and k_lstm_gs (k_gru_gs's structure for one-layer LSTMs: unit-major tiles, a main wave per 16 units, the chain on helper
// waves): LSTM-64 265 us per round of 4096 streams — ahead of k_quad from 1025 streams on (289 us at 2048) and of k_mfma_ls1
// throughout (308 / 616 / 1 234 us at 4096 / 8192 / 16 384 against 265 / 528 / 1 057); LSTM-40 265 us, between k_quad (223 us at
// 2048) and k_mfma_ls1 (two workgroups per CU: 418 us at 8192 against 527) — profiles/r04_ls1_ab.txt. AIDAX_LSTM_GS=0 / 1: never / wherever it serves
// (LSTM-80 is served too — five main waves, two helpers — but only 4 % ahead of k_mfma_ls1, 483 against 503 us per 4096 streams: not in the rule).
bool lstm_gs_pays(int cell, int hidden, uint32_t n, int cus)
{
    // (read on every call — the worker thread's, at model load: tests switch forms within one process)
    const int forced = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_LSTM_GS"); return !e ? -1 : e[0] != '0' ? 1 : 0; }();
    const bool f32 = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_GRU_GM"); return e && e[0] == 'f'; }();
    if (cell != AIDAX_CELL_LSTM || (hidden != 40 && hidden != 64 && hidden != 80) || forced == 0 || f32) return false;
    if (forced == 1) return true;
    if (cus <= 0) return false;
    const uint32_t groups = (n + kMfmaStreams - 1) / kMfmaStreams, c = static_cast<uint32_t>(cus);
    return hidden == 64 ? groups * 4 > c : hidden == 80 ? false : groups * 2 > c && groups <= c;
}

ManyForm many_streams_form(int cell, int hidden, uint32_t n, int cus, uint32_t max_frames)
{
    if (lstm_gs_pays(cell, hidden, n, cus)) return MANY_MFMA;
    const bool lstm = cell == AIDAX_CELL_LSTM;
    const uint32_t groups = (n + kMfmaStreams - 1) / kMfmaStreams;
    const bool ls1_off = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_LS1"); return e && e[0] == '0'; }();
    if (!ls1_off && lone_split_pays(cell, hidden, n, cus, max_frames)) return MANY_MFMA;
    // Round 5: LSTM-32 between one and one and a half rounds of stream groups (4097 .. 6144 streams on 256 CUs) was k_quad's by the
    // rule below — which nothing measured there ever supported: k_nn (the split form) is 5 .. 16 % ahead of it at every block length
    // (5120 streams: 77 / 135 / 249 us at 64 / 128 / 256 frames against 84 / 146 / 271), and k_mfma_ls1 takes over above (lone_split_pays).
    if (lstm && hidden == 32 && cus > 0 && groups > static_cast<uint32_t>(cus) && groups * 2 <= static_cast<uint32_t>(cus) * 3) return MANY_NONE;
    const bool full_round = cus > 0 && groups <= static_cast<uint32_t>(cus) && groups * 8 > static_cast<uint32_t>(cus) * 7;
    if (full_round && (hidden == 32 || hidden == 64 || (hidden == 40 && lstm))) return MANY_MFMA;
    // One-layer GRUs of 40 (run as 48) / 64 units have k_gru_gm (gate-major tiles: three quarters of the matrix-core work,
    // the whole run() in one launch; up to two workgroups per CU): GRU-64 326 us flat up to 4096 streams, 595 us at 8192,
    // 1 140 us at 16384 against 257 / 408 / 859 us for k_nn at 2048 / 3072 / 8192; GRU-40 277-287 us up to 4096 against
    // 300 (k_nn) / 278 (k_quad) at 3072 and 303 (k_quad) at 4096, 492 / 562 at 8192.
    // Round 4: with the recurrent product on the bf16 matrix pipe (k_gru_gs) the kernel costs 190 us (GRU-64) / 186 us (GRU-40) per
    // 256-frame block whatever the stream count up to a full round, and the crossovers moved (scratch/gs_threshold.py,
    // profiles/r04_gs_threshold.txt): GRU-64 wins from 256 streams on (k_quad 197 us at 256 - 1024, k_nn 225 at 1536 - 2048);
    // GRU-40 from the point where k_nn needs a second round of waves (2048 streams: 174 us; 2560: 279). AIDAX_GRU_GM=f32 (the
    // fp32 MFMA kernel, 305 us) keeps round 3's thresholds.
    const bool gm_f32 = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_GRU_GM"); return e && e[0] == 'f'; }();
    // Round 5: GRU-64 is k_gru_gs's at EVERY pool size, the one-stream pool of an LV2 instance included — 49.9 / 93.1 / 179 us per block
    // of 64 / 128 / 256 frames at one stream against 55.4 / 99.1 / 187 on k_quad, 54 / 99 / 188 against 64 / 108 / 197 at 16 .. 192 streams
    // (the 256-stream threshold of round 4 was measured at 256 frames only, where the gap is 4 %; at a 64-frame period it is 19 %).
    // Round 6 (advisor): that evidence is blocks of 64 frames and more; a pool created for shorter blocks than any measured (hub mode's
    // four-frame placeholder pool) keeps round 4's threshold — a sixteenth of a round of stream groups (256 streams on 256 CUs).
    if (!lstm && cus > 0 && !gm_f32 && ((hidden == 64 && (max_frames >= 64 || groups * 16 >= static_cast<uint32_t>(cus))) || (hidden == 40 && groups * 2 > static_cast<uint32_t>(cus))))
        return MANY_MFMA;
    if (!lstm && cus > 0 && ((hidden == 64 && groups * 8 >= static_cast<uint32_t>(cus) * 5) || (hidden == 40 && groups * 8 >= static_cast<uint32_t>(cus) * 7)))
        return MANY_MFMA;
    // What is left of the table, in ROUNDS of stream groups like the rules above (round 6: these were stream counts measured on 256 CUs —
    // 4096 streams = one round of sixteen-stream workgroups there; a partition with 32 or 64 CUs reaches its round at 512 or 1024
    // streams). A device whose CU count is unknown (cus <= 0) is taken for the one the table was measured on.
    const uint32_t round = static_cast<uint32_t>(kMfmaStreams) * (cus > 0 ? static_cast<uint32_t>(cus) : 256u);      // streams of one round of workgroups
    if (hidden <= 32) return n >= round ? MANY_QUAD : MANY_NONE;
    if (hidden == 40) return n >= 2 * round ? MANY_MFMA : n >= (lstm ? round / 8 : round) ? MANY_QUAD : MANY_NONE;
    if (hidden == 64 && !lstm) return n >= 4 * round ? MANY_MFMA : n <= round / 4 ? MANY_QUAD : MANY_NONE;
    // LSTM-64, LSTM-80, GRU-80: their one-wave kernels hold 250-500 weight registers; k_quad is 1.35-1.75x
    // faster already at 64 streams, k_mfma takes over from a full round of stream groups
    return n >= round ? MANY_MFMA : MANY_QUAD;
}

// k_mfma_lp needs every workgroup of its grid resident at once, which the pool can promise only for ONE grid on the
// device: one pool per device holds the right to use the kernel (a pool's staged model shares its own pool's hold), the
// others serve their stacked models with k_mfma. Process-wide; another process on the same GPU is beyond its reach
// and ends in a reported give-up (aidax_mfmalp.hip).
// a flag written on the launch path and read by other threads (aidax_pool_kernel_name, lp_in_use from the worker): an atomic that
// a struct assignment of its owner (a model swap is one) may copy
struct RelaxedFlag {
    std::atomic<bool> v{false};
    RelaxedFlag() = default;
    RelaxedFlag(const RelaxedFlag& o) : v(o.v.load(std::memory_order_relaxed)) {}
    RelaxedFlag& operator=(const RelaxedFlag& o) { v.store(o.v.load(std::memory_order_relaxed), std::memory_order_relaxed); return *this; }
    RelaxedFlag& operator=(bool b) { v.store(b, std::memory_order_relaxed); return *this; }
    operator bool() const { return v.load(std::memory_order_relaxed); }
};

struct LpGate {
    std::mutex mu;
    // `off`: the owner's "I have stopped using the kernel" flag (a pool whose hand-over gave up serves its model with
    // k_mfma from then on): a hold whose owner has raised it is void, the next pool that asks takes the device over
    struct Hold { const void* owner = nullptr; int refs = 0; const std::atomic<bool>* off = nullptr; } dev[64];
    bool acquire(int device, const void* owner, const std::atomic<bool>* off = nullptr)
    {
        if (device < 0 || device >= 64) return false;
        std::lock_guard<std::mutex> g(mu);
        Hold& h = dev[device];
        if (h.refs != 0 && h.owner != owner) {
            if (!(h.off && h.off->load(std::memory_order_relaxed))) return false;
            h.refs = 0;                                    // (the old owner's later release() no longer matches and is ignored)
        }
        h.owner = owner;
        h.off = off;
        ++h.refs;
        return true;
    }
    void release(int device, const void* owner)
    {
        if (device < 0 || device >= 64) return;
        std::lock_guard<std::mutex> g(mu);
        Hold& h = dev[device];
        if (h.owner == owner && h.refs > 0 && --h.refs == 0) { h.owner = nullptr; h.off = nullptr; }
    }
};
LpGate& lp_gate() { static LpGate g; return g; }

// AIDAX_KEEP_WARM_US=<period in us> (off unless set): a host calls run() once per audio period (rt-neural-generic.cpp:484) — 1.3 to 5.3 ms
// apart — and a GPU that has idled for that long serves the next launch slower than one that is kept busy (bench.py realtime_paced: the
// same pass 8 - 13 % longer, the call's tail several times longer). With the variable set, ONE thread per device launches an empty grid
// (a wave per CU) on a stream of the lowest priority every <period>, while any pool on that device exists. Never on the audio thread;
// what it buys on a box is measured by bench.py (realtime_paced.keep_warm) and stated in INTEGRATION.md §3.
struct KeepWarm {
    std::mutex mu;
    struct Dev { int refs = 0; std::thread th; std::atomic<bool> stop{false}; } dev[64];
    ~KeepWarm()
    {
        for (Dev& d : dev) { d.stop.store(true); if (d.th.joinable()) d.th.detach(); }      // (a pool that outlived main(): no join at exit)
    }
    static long period_us()
    {
        const char* e = std::getenv("AIDAX_KEEP_WARM_US");
        const long v = e ? std::strtol(e, nullptr, 10) : 0;
        return v >= 50 && v <= 1000000 ? v : 0;
    }
    bool acquire(int device)
    {
        const long us = period_us();
        if (us == 0 || device < 0 || device >= 64) return false;
        std::lock_guard<std::mutex> g(mu);
        Dev& d = dev[device];
        if (d.refs++ == 0) {
            d.stop.store(false);
            d.th = std::thread([device, us, &d] {
                if (hipSetDevice(device) != hipSuccess) return;
                int least = 0, greatest = 0, cus = 0;
                hipStream_t q = nullptr;
                (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                if (hipStreamCreateWithPriority(&q, hipStreamNonBlocking, least) != hipSuccess) return;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
                auto next = std::chrono::steady_clock::now();
                while (!d.stop.load(std::memory_order_relaxed)) {
                    (void)launch_keep_warm_kernel(cus > 0 ? cus : 1, q);
                    next += std::chrono::microseconds(us);
                    const auto now = std::chrono::steady_clock::now();
                    if (next < now) next = now;
                    std::this_thread::sleep_until(next);
                }
                (void)hipStreamSynchronize(q);
                (void)hipStreamDestroy(q);
            });
        }
        return true;
    }
    void release(int device)
    {
        std::thread done;
        {
            std::lock_guard<std::mutex> g(mu);
            Dev& d = dev[device];
            if (d.refs > 0 && --d.refs == 0) { d.stop.store(true); done = std::move(d.th); }
        }
        if (done.joinable()) done.join();
    }
};
KeepWarm& keep_warm() { static KeepWarm k; return k; }

struct HipFail : std::runtime_error { using std::runtime_error::runtime_error; };

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        std::ostringstream os;
        os << what << ": " << hipGetErrorString(e);
        throw HipFail(os.str());
    }
}
#define HIP_TRY(x) hip_check((x), #x)

template <class F>
int guarded(F&& f)
{
    try {
        return f();
    } catch (const HipFail& e) {
        return fail(AIDAX_ERR_DEVICE, e.what());
    } catch (const std::exception& e) {
        return fail(AIDAX_ERR_STATE, e.what());
    } catch (...) {
        return fail(AIDAX_ERR_STATE, "unknown failure");
    }
}

// The device side of one loaded model: what the reference keeps in a DynamicModel (rt-neural-generic.h:115-129)
// minus the per-stream members. A pool plays `cur`; aidax_pool_prepare_model builds the next one on the worker
// thread and aidax_pool_commit_model swaps the two on the audio thread (plain struct assignment).
struct ModelSlot {
    bool has_model = false;
    enum Kind { TABLE = 0, STACK = 1, CONV = 2, MFMA = 3, QUAD = 4 } kind = TABLE;
    QuadDesc qdesc{};
    StackDesc sdesc{};
    MfmaDesc mdesc{};
    ConvDesc cdesc{};
    bool conv_mfma = false;          // conv stacks on the matrix cores (blocks of <= 256 frames), else k_conv
    bool conv_ms = false;            // ... as bf16 term products (k_conv_ms: the stacks conv_ms_shape_ok admits; its own history layout)
    bool conv_fused = false;         // ... with the DSP chain inside the same launch (AIDAX_CONV_FUSED=0: packed k_chain launches around it)
    const KernelEntry* kernel = nullptr;
    int cell = 0, input_size = 1, input_skip = 0, hidden = 0;
    float in_gain = 1.f, out_gain = 1.f, model_sr = 48000.f;
    uint32_t nn_stride = 0;
    int pipe_capacity = 0;           // streams the 3-wave pipeline keeps resident at once (0 = never use it)
    bool split_pays = false;         // the lean recurrent kernel keeps the occupancy of the one-wave kernel
    float* d_wq4 = nullptr;          // k_lstm_q4's weight record (LSTM-32 snapshot models), next to the table kernels' d_wpack
    float* d_wpack = nullptr;        // weights in the kernel's layout
    float* d_nn = nullptr;           // recurrent state [n_streams][nn_stride]
    float* d_ring = nullptr;         // k_mfma_lp: h of layer l-1 on its way to layer l, per stream group
    uint32_t* d_counters = nullptr;  // k_mfma_lp: frames produced / consumed per (group, layer boundary)
    mutable const void* lp_owner = nullptr;  // the pool whose hold on the device's LpGate this slot shares (d_ring != nullptr); given up on the launch path when the grid is refused
    bool lp_fused = false;           // k_mfma_lp runs the DSP chain too (one-layer models, helper waves): one launch per block
    uint32_t lp_round_streams = 0;   // k_mfma_ls on a pool larger than one resident grid: streams per launch (a pass is several launches over stream ranges); 0: one launch
    mutable RelaxedFlag lp_refused;  // the runtime refused this model's chained grid (cooperative launch: not co-resident on this device): k_mfma serves it — this
                                     // model only; another model of the pool, with a smaller grid, may still fit (set on the launch path, hence mutable)
    int lp_split = 0;                // stacked models: k_mfma_ls serves the passes (contractions as bf16 term products of split operands): 6 or 9 products; 0: k_mfma_lp (fp32 MFMAs)
    bool gru_gm = false;             // one-layer GRU: k_gru_gm (gate-major tiles, the whole run() in one launch) serves the passes
    int lstm_gs = 0;                 // one-layer LSTM-40 / 64 on k_lstm_gs (k_gru_gs's structure): 6 or 9 term products; 0: not
    int gru_gs = 0;                  // ... as k_gru_gs (recurrent product on the bf16 matrix pipe, operands split into three bf16 terms): 6 or 9 term products; 0: the fp32 kernel

    float p_den() const { return 0.1f * model_sr; }      // LinearValueSmoother tau * sampleRate (:1053-1054)
};

constexpr int kCtlRing = 4;                              // pinned snapshots of the control records in flight
constexpr size_t kStagingLimit = size_t(64) << 20;      // pinned staging per direction for aidax_pool_process
constexpr size_t kZeroCopyLimit = size_t(64) << 10;     // blocks up to this size are read / written by the kernels in place in pinned host memory

}  // namespace
}  // namespace aidax

// A model on its way into (or out of) a pool. Built by aidax_pool_prepare_model on the worker thread; after
// aidax_pool_commit_model the same object carries what the swap retired, for aidax_staged_free on the worker.
struct aidax_staged {
    int device = 0;
    uint32_t n_streams = 0;
    aidax::ModelSlot slot;
    StreamState* d_pst = nullptr;    // per-stream DynamicModel members of the new model (PARAM smoothers, paramFirstRun), installed by the commit
    hipEvent_t fence = nullptr;      // recorded by the commit: everything that may still touch the retired buffers precedes it
    bool fenced = false;
};

struct aidax_pool {
    int device = 0;
    uint32_t n_streams = 0, max_frames = 0;
    double host_sr = 48000.0;
    float gain_coef = 0.f;
    hipStream_t q = nullptr;         // the audio side's stream
    hipStream_t wq = nullptr;        // the worker side's stream (prepare)
    hipEvent_t ev_x = nullptr;       // edge between two streams that carry passes one after the other
    hipEvent_t ev_adopt = nullptr;   // "everything this pool has issued so far": what a stream of ANOTHER pool waits for before it adopts one of ours
    hipStream_t last_stream = nullptr;

    StreamCtl* d_ctl = nullptr;
    StreamState* d_st = nullptr;
    float* d_in = nullptr;           // staging for the host-buffer entry point
    float* d_out = nullptr;
    float* h_in = nullptr;           // pinned host staging (nullptr: blocks too large, pageable copies)
    float* h_out = nullptr;
    float* hd_in = nullptr;          // device view of h_in / h_out (zero-copy passes)
    float* hd_out = nullptr;
    bool zero_copy = false;
    // completion word of a pass in pinned host memory: the stream writes the pass number behind its last launch and the
    // caller of aidax_pool_process polls it — no interrupt, no wake-up of a blocked thread (small pools: the plugin case)
    uint32_t* h_done = nullptr;
    uint32_t* hd_done = nullptr;
    uint32_t done_seq = 0;
    bool spin_wait = false;
    bool kernel_word = true;         // a one-workgroup pass (k_*_pipe / k_*_pipe4 of a one-stream pool) writes the completion word itself, behind its block:
                                     // no packet follows the pass on its queue — 1.7 us of the LV2 instance's 23 us round trip at 64 frames
                                     // (profiles/r06_host_pipeline.txt). AIDAX_KERNEL_WORD=0: the queue writes it (hipStreamWriteValue32), as in round 5
    bool spin_collect = true;        // aidax_pool_collect polls the download's event instead of sleeping on it (AIDAX_SPIN_WAIT=0: off)
    bool keep_warm = false;          // this pool holds a reference on its device's keep-warm thread (AIDAX_KEEP_WARM_US)
    // k_mfma_lp: a word in pinned host memory that a workgroup bumps when a layer hand-over timed out. The pass that did
    // so is wrong; whoever notices reports it, and the pool serves the model with k_mfma from then on (lp_off).
    uint32_t* h_lp_fault = nullptr;
    uint32_t* hd_lp_fault = nullptr;
    mutable std::atomic<bool> lp_off{false};      // (mutable: set from the launch path, a const member function)
    std::atomic<uint32_t> lp_faults{0};
    mutable std::atomic<uint32_t> lp_refusals{0}; // chained grids the runtime refused (the text is in aidax_last_error(); kernel_name says k_mfma from then on)
    bool take_lp_fault()
    {
        if (!h_lp_fault) return false;
        // read-and-clear in ONE atomic step: workgroups bump the word with system-scope atomic adds while we look, and a
        // give-up that landed between a read and a separate store of 0 would go unreported
        if (__atomic_load_n(h_lp_fault, __ATOMIC_RELAXED) == 0) return false;
        if (__atomic_exchange_n(h_lp_fault, 0u, __ATOMIC_ACQ_REL) == 0) return false;
        lp_off.store(true, std::memory_order_relaxed);
        lp_faults.fetch_add(1, std::memory_order_relaxed);
        return true;
    }

    // aidax_pool_submit / aidax_pool_collect: kPipeSets staging sets and two copy streams, so that the upload of the blocks behind
    // block k and the download of the blocks in front of it run under the pass of block k (allocated by the first submit).
    // THREE sets since round 6: with two, submit(k) has to wait for collect(k - 2), which returns a cross-stream hop and a download
    // after pass k - 2 has ended; the upload of k and its own hop then end ~110 us after that — later than pass k - 1 (65 us for a
    // cfg2 block) does, and the GPU idles for the difference every block (88 us per block predicted, 83.8 measured in round 3).
    // A third set takes the host's round trip off the GPU's critical path: the pass is what is left.
    static constexpr int kPipeSets = 3;
    struct Pipeline {
        bool ready = false;
        float* h_in[kPipeSets] = {}; float* h_out[kPipeSets] = {}; float* d_in[kPipeSets] = {}; float* d_out[kPipeSets] = {};
        hipEvent_t ev_up[kPipeSets] = {}, ev_pass[kPipeSets] = {}, ev_down[kPipeSets] = {};
        hipStream_t q_up = nullptr, q_down = nullptr;
        uint32_t frames[kPipeSets] = {};
        float* direct_out[kPipeSets] = {};      // the block's download went straight to this caller buffer (registered): collect() only waits
        uint64_t submitted = 0, collected = 0;
    } pipe;
    // aidax_pool_register_host: page-locked ranges of the caller's memory (copies to and from them need no staging)
    struct HostRange { char* base; size_t bytes; };
    std::vector<HostRange> host_ranges;
    bool host_registered(const void* ptr, size_t bytes) const
    {
        const char* c = static_cast<const char*>(ptr);
        for (const HostRange& r : host_ranges)
            if (c >= r.base && c + bytes <= r.base + r.bytes) return true;
        return false;
    }

    std::vector<aidax_controls> controls;
    std::vector<uint8_t> loading;
    std::vector<uint8_t> forced_off;                     // the hub parks a stream (raw copy, state does not move) without touching its controls
    std::vector<StreamCtl> h_ctl;
    uint32_t dirty_lo = 1, dirty_hi = 0;                 // control records to upload: [lo, hi], empty when lo > hi
    StreamCtl* ctl_ring[kCtlRing] = {};
    hipEvent_t ctl_ev[kCtlRing] = {};
    bool ctl_used[kCtlRing] = {};
    int ctl_next = 0;

    ModelSlot cur;
    int tune = 0;                    // AIDAX_TUNE (measurement switches, see LaunchArgs)
    int cus = 0;                     // compute units of the device (form selection)
    int force_form = 0;              // AIDAX_KERNEL=wave|pipe|split|valu|mfma|quad overrides the heuristic (A/B testing)

    // Form of a MODE_CHAIN pass of a TABLE slot: 0 one wave per stream, 1 three-wave pipeline (all streams resident at
    // once: latency-bound regime), 2 split launches with packed chains (many streams: issue-bound regime), 3 four
    // streams per workgroup with the cell on the matrix cores (k_lstm_q4: about four streams per CU or fewer)
    int chain_form(const ModelSlot& m) const
    {
        if (m.kind != ModelSlot::TABLE && m.has_model) return 0;
        if (m.has_model && m.d_wq4 && force_form == 7) return 3;      // opt-in only (AIDAX_KERNEL=q4): measured slower than the pipeline, see aidax_q4.hip
        if (force_form == 1) return 0;
        if (force_form == 2) return (m.has_model && m.kernel && m.kernel->fn_pipe) ? 1 : 0;
        const bool packed_fits = chain_lds_bytes(max_frames) <= kChainLdsLimit;      // packed chains keep 8 blocks in LDS
        if (force_form == 3) return (packed_fits && (!m.has_model || (m.kernel && m.kernel->fn_nn))) ? 2 : 0;
        if (use_pipe(m)) return 1;
        if (n_streams < 64 || !packed_fits) return 0;
        return (!m.has_model || m.split_pays) ? 2 : 0;
    }
    // Models of the reference's table on the matrix-core kernels (many_streams_form); AIDAX_KERNEL=quad|mfma force one.
    bool quad_for_table_model(int cell_kind, int hidden_units) const
    {
        if (force_form == 6) return true;
        return force_form == 0 && many_streams_form(cell_kind, hidden_units, n_streams, cus, max_frames) == MANY_QUAD;
    }
    bool mfma_for_table_model(int cell_kind, int hidden_units) const
    {
        if (force_form == 5) return true;
        return force_form == 0 && many_streams_form(cell_kind, hidden_units, n_streams, cus, max_frames) == MANY_MFMA;
    }
    bool use_pipe(const ModelSlot& m) const
    {
        if (!m.has_model || !m.kernel || !m.kernel->fn_pipe || m.kind != ModelSlot::TABLE) return false;
        if (force_form == 1 || force_form == 3) return false;
        if (force_form == 2) return true;
        return static_cast<int>(n_streams) <= m.pipe_capacity;
    }
    // does a MODE_CHAIN pass of this slot consist of one launch (it may then read and write host memory in place)?
    bool single_launch(const ModelSlot& m) const
    {
        if (!m.has_model) return chain_form(m) != 2;
        switch (m.kind) {
        case ModelSlot::TABLE: return chain_form(m) != 2;
        case ModelSlot::STACK: return true;
        case ModelSlot::CONV: return !m.conv_mfma || m.conv_fused;
        default: return false;
        }
    }

    bool lp_in_use(const ModelSlot& m) const { return m.d_ring != nullptr && !m.lp_refused && !(m.mdesc.n_layers >= 2 && lp_off.load(std::memory_order_relaxed)); }

    static size_t lds_bytes(const ModelSlot& m, uint32_t n_frames)
    {
        return (static_cast<size_t>((n_frames + 3) & ~3u) + static_cast<size_t>(m.hidden > 0 ? m.hidden + 4 : 4)) * sizeof(float);   // block + h row + spare slot
    }

    // Does a MODE_CHAIN pass of n frames go through the four-streams-per-workgroup pipeline (k_*_pipe4: 61.9 us per cfg2 block against
    // k_lstm_pipe's 63.8, the small cells 9 - 19 % ahead, conditioned models 9 - 25 % at every cell up to 32: profiles/r06_cfg2_pipe4.txt, r06_pipe4_cells.txt)? Whole 16-frame tiles, at least one full
    // workgroup of four streams (a conditioned model: any pool), every workgroup on a CU of its own; k_*_pipe serves every other pass on the same
    // state (AIDAX_PIPE4=0, test build: all of them).
    // (a plain model below one full workgroup: level at 256 frames, 0.6 - 1.0 us behind at 64 — the helper wave's longer prologue; a conditioned
    // model is 11 - 20 % ahead also as a pool of ONE stream, the LV2 instance's: profiles/r06_pipe4_cells.txt. AIDAX_PIPE4_MIN, test build: the measurement's switch)
    static uint32_t p4_min_streams(int input_size) { const char* e = AIDAX_HOOK_ENV("AIDAX_PIPE4_MIN"); return e ? (uint32_t)atoi(e) : input_size > 1 ? 1u : 4u; }
    bool pipe4_serves(const ModelSlot& m, uint32_t n, int input_size) const
    {
        const bool off = [] { const char* e = AIDAX_HOOK_ENV("AIDAX_PIPE4"); return e && e[0] == '0'; }();      // (read per call: tests switch forms within one process)
        if (off || force_form != 0 || !m.kernel || !(input_size > 1 ? m.kernel->fn_pipe4c : m.kernel->fn_pipe4) || input_size < 1 || input_size > 3 || n == 0 || n % 16u || n_streams < p4_min_streams(input_size)) return false;
        return cus > 0 && (n_streams + 3u) / 4u <= static_cast<uint32_t>(cus) && pipe4_lds_bytes(m.hidden, n, input_size) <= 160 * 1024;
    }
    mutable hipEvent_t pass_done = nullptr;  // pool_submit_impl -> launch(): the event that marks this pass's end, for a launch that can carry it
    mutable bool pass_done_taken = false;
    mutable uint32_t* pass_word = nullptr;   // aidax_pool_process -> launch(): the completion word and the number to write, for a pass of one workgroup
    mutable uint32_t pass_seq = 0;
    mutable bool pass_word_taken = false;
    void mark_dirty(uint32_t lo, uint32_t hi)
    {
        if (dirty_lo > dirty_hi) { dirty_lo = lo; dirty_hi = hi; }
        else { dirty_lo = std::min(dirty_lo, lo); dirty_hi = std::max(dirty_hi, hi); }
    }
    void refresh_ctl(uint32_t s)
    {
        build_stream_ctl(controls[s], host_sr, cur.has_model, loading[s] != 0, gain_coef, cur.p_den(), &h_ctl[s]);
        if (forced_off[s]) h_ctl[s].flags &= ~static_cast<uint32_t>(CTL_ENABLED);
        mark_dirty(s, s);
    }
    void refresh_all()
    {
        // identical controls are the common case: design once, then copy
        for (uint32_t s = 0; s < n_streams; ++s) {
            if (s > 0 && std::memcmp(&controls[s], &controls[s - 1], sizeof(aidax_controls)) == 0 &&
                loading[s] == loading[s - 1]) {
                h_ctl[s] = h_ctl[s - 1];
            } else {
                build_stream_ctl(controls[s], host_sr, cur.has_model, loading[s] != 0, gain_coef, cur.p_den(), &h_ctl[s]);
            }
        }
        for (uint32_t s = 0; s < n_streams; ++s)
            if (forced_off[s]) h_ctl[s].flags &= ~static_cast<uint32_t>(CTL_ENABLED);
        mark_dirty(0, n_streams - 1);
    }
    // the fields of a control record that depend on the model and on `loading`, without redesigning the filters
    void patch_model_fields(uint32_t s)
    {
        StreamCtl& o = h_ctl[s];
        const aidax_controls& c = controls[s];
        o.master_target = loading[s] ? 0.f : db_to_coeff(c.master_db);                   // :490, :654
        o.p_den = cur.p_den();
        if (cur.has_model && !(c.net_bypass > 0.5f)) o.flags |= CTL_NET_ON;              // :631-632
        else o.flags &= ~static_cast<uint32_t>(CTL_NET_ON);
    }
    // Upload the changed control records, stream-ordered with the pass that follows. The records leave from a
    // ring of pinned snapshots, so the copy is asynchronous and a later set_controls cannot overtake it.
    void flush_ctl(hipStream_t s)
    {
        if (dirty_lo > dirty_hi) return;
        const int k = ctl_next;
        if (ctl_used[k] && hipEventQuery(ctl_ev[k]) != hipSuccess) HIP_TRY(hipEventSynchronize(ctl_ev[k]));   // four uploads behind: not seen in practice
        const size_t cnt = static_cast<size_t>(dirty_hi - dirty_lo) + 1;
        std::memcpy(ctl_ring[k] + dirty_lo, h_ctl.data() + dirty_lo, cnt * sizeof(StreamCtl));
        HIP_TRY(hipMemcpyAsync(d_ctl + dirty_lo, ctl_ring[k] + dirty_lo, cnt * sizeof(StreamCtl), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(ctl_ev[k], s));
        ctl_used[k] = true;
        ctl_next = (k + 1) % kCtlRing;
        dirty_lo = 1; dirty_hi = 0;
    }
    // Passes and control pokes are issued in program order by the audio side but may sit on two streams (the pool's
    // own and one handed to aidax_pool_process_device): moving from one to the other puts an event edge between them.
    void enter_stream(hipStream_t s)
    {
        if (last_stream && last_stream != s) {
            HIP_TRY(hipEventRecord(ev_x, last_stream));
            HIP_TRY(hipStreamWaitEvent(s, ev_x, 0));
        }
        last_stream = s;
    }
    LaunchArgs args(const ModelSlot& m, StreamState* st, const float* in, float* out, uint32_t n_frames, int mode) const
    {
        LaunchArgs a{};
        a.ctl = d_ctl; a.st = st; a.nn = m.d_nn; a.wpack = m.d_wpack;
        a.in = in; a.out = out;
        a.n_streams = n_streams; a.n_frames = n_frames; a.nn_stride = m.nn_stride;
        a.mode = mode; a.input_size = m.input_size; a.input_skip = m.input_skip;
        a.in_gain = m.in_gain; a.out_gain = m.out_gain;
        a.tune = tune;
        return a;
    }
    // frames one launch of the extension kernels may carry (their LDS planes grow with n)
    uint32_t ext_chunk() const { return max_frames < 256 ? max_frames : 256; }
    uint32_t launch_chunk(const ModelSlot& m, uint32_t n) const
    {
        // (MFMA: k_mfma takes any length, but a 2048-frame warm-up launch would hold its CUs for ~10 ms next to the passes)
        return (m.kind == ModelSlot::STACK || m.kind == ModelSlot::CONV || m.kind == ModelSlot::MFMA) ? std::min(ext_chunk(), n) : n;
    }
    hipError_t launch(const ModelSlot& m, const LaunchArgs& a, hipStream_t s) const
    {
        if (m.has_model && m.kind == ModelSlot::MFMA) {
            // split form around the matrix-core kernel: packed chains in -> out, applyModel in place, packed chains
            // k_mfma_lp only for the passes themselves: warm-ups (worker stream, next to the passes) and the bare-model
            // modes run on k_mfma, which leaves bit-identical state — two of those grids must never be in flight together
            // A chained grid the runtime refuses (cooperative launch: it cannot be co-resident on this device — a partitioned
            // GPU, a pool forced onto the kernel) is no error of the pass: the model is served by k_mfma from here on
            auto refused = [&](hipError_t e) {
                if (e != hipErrorCooperativeLaunchTooLarge) return false;
                (void)hipGetLastError();
                m.lp_refused = true;                          // this model's grid; a pool-wide lp_off is for give-ups (co-tenants)
                // ... and this slot's share of the pool's hold on the device's gate goes back: a pool that cannot use the chained kernels
                // must not keep every other pool of the device off them (advisor, round 5)
                if (m.lp_owner) { lp_gate().release(device, m.lp_owner); m.lp_owner = nullptr; }
                lp_refusals.fetch_add(1, std::memory_order_relaxed);
                set_error("a chained grid cannot be co-resident on this device (cooperative launch refused): the model is served by k_mfma");
                return true;
            };
            // k_mfma_ls over the pass's streams: one launch, or one per range of lp_round_streams streams (every range a resident grid)
            auto launch_ls = [&](bool fused) {
                if (!m.lp_round_streams || a.n_streams <= m.lp_round_streams)
                    return launch_mfma_ls_kernel(a, m.mdesc, m.d_ring, m.d_counters, hd_lp_fault, m.lp_split, s, fused);
                for (uint32_t s0 = 0; s0 < a.n_streams; s0 += m.lp_round_streams) {
                    LaunchArgs b = a;
                    b.n_streams = std::min(m.lp_round_streams, a.n_streams - s0);
                    b.ctl = a.ctl + s0; b.st = a.st + s0; b.nn = a.nn + static_cast<size_t>(s0) * a.nn_stride;
                    if (a.in) b.in = a.in + static_cast<size_t>(s0) * a.n_frames;
                    b.out = a.out + static_cast<size_t>(s0) * a.n_frames;
                    const hipError_t e = launch_mfma_ls_kernel(b, m.mdesc, m.d_ring, m.d_counters, hd_lp_fault, m.lp_split, s, fused);
                    if (e != hipSuccess) return e;      // (a refusal can only come from the first range: the ranges are one size or smaller)
                }
                return hipSuccess;
            };
            auto model_kernel = [&]() {
                if (lp_in_use(m) && a.mode == MODE_CHAIN) {
                    const hipError_t e = m.lp_split ? launch_ls(false) : launch_mfma_lp_kernel(a, m.mdesc, m.d_ring, m.d_counters, hd_lp_fault, s);
                    if (!refused(e)) return e;
                }
                return launch_mfma_kernel(a, m.mdesc, s);
            };
            if (a.mode != MODE_CHAIN) return model_kernel();
            if (m.lstm_gs && a.n_frames != 0) return launch_lstm_gs_kernel(a, m.mdesc, m.lstm_gs, s);
            if (m.gru_gm && a.n_frames != 0) return m.gru_gs ? launch_gru_gs_kernel(a, m.mdesc, m.gru_gs, s) : launch_gru_gm_kernel(a, m.mdesc, s);
            if (m.lp_fused && lp_in_use(m) && a.n_frames != 0) {
                const hipError_t e = m.lp_split ? launch_ls(true) : launch_mfma_lp_kernel(a, m.mdesc, m.d_ring, m.d_counters, hd_lp_fault, s, true);
                if (!refused(e)) return e;
            }
            hipError_t e = launch_chain_pass(true, a, s);
            if (e == hipSuccess && a.n_frames != 0) e = model_kernel();
            if (e == hipSuccess) e = launch_chain_pass(false, a, s);
            return e;
        }
        if (m.has_model && m.kind == ModelSlot::QUAD) {
            if (a.mode != MODE_CHAIN) return launch_quad_kernel(m.cell, m.hidden, a, m.qdesc, s);
            hipError_t e = launch_chain_pass(true, a, s);
            if (e == hipSuccess && a.n_frames != 0) e = launch_quad_kernel(m.cell, m.hidden, a, m.qdesc, s);
            if (e == hipSuccess) e = launch_chain_pass(false, a, s);
            return e;
        }
        if (m.has_model && m.kind == ModelSlot::STACK) return launch_stack_kernel(a, m.sdesc, s);
        if (m.has_model && m.kind == ModelSlot::CONV && m.conv_mfma) {
            auto conv_launch = [&](const LaunchArgs& b, bool fused) {
                return m.conv_ms ? launch_conv_ms_kernel(b, m.cdesc, fused, s) : launch_conv_mfma_kernel(b, m.cdesc, fused, s);
            };
            if (a.mode != MODE_CHAIN) return conv_launch(a, false);
            if (m.conv_fused) {
                // the whole run() in one launch; a block longer than the kernel's 256 frames goes through in time slices,
                // each the whole run() of its slice (the rows keep the block's pitch)
                const uint32_t chunk = ext_chunk();
                if (a.n_frames <= chunk) return conv_launch(a, true);
                hipError_t e = hipSuccess;
                for (uint32_t done = 0; done < a.n_frames && e == hipSuccess; done += chunk) {
                    LaunchArgs b = a;
                    b.in = a.in + done; b.out = a.out + done;
                    b.n_frames = std::min(chunk, a.n_frames - done);
                    b.row_stride = a.n_frames;
                    e = conv_launch(b, true);
                }
                return e;
            }
            hipError_t e = launch_chain_pass(true, a, s);
            if (e == hipSuccess && a.n_frames != 0) e = conv_launch(a, false);
            if (e == hipSuccess) e = launch_chain_pass(false, a, s);
            return e;
        }
        if (m.has_model && m.kind == ModelSlot::CONV) return launch_conv_kernel(a, m.cdesc, s);
        const int form = a.mode == MODE_CHAIN ? chain_form(m) : 0;
        if (form == 3) {
            LaunchArgs b = a;
            b.wpack = m.d_wq4;
            return launch_q4_kernel(m.hidden, b, s);
        }
        if (form == 1) {
            // (the pipelined host path hands the pass its "pass done" event: riding on the dispatch it saves the marker packet behind the kernel)
            hipEvent_t done = pass_done;
            pass_done = nullptr;
            pass_done_taken = done != nullptr;
            LaunchArgs b = a;
            if (pass_word && a.n_streams == 1 && a.n_frames != 0) {      // ... or the blocking path's completion word, written by the one workgroup itself
                b.done_word = pass_word; b.done_seq = pass_seq;
                pass_word = nullptr;
                pass_word_taken = true;
            }
            if (!pipe4_serves(m, a.n_frames, a.input_size)) return launch_pipe_kernel(m.kernel, b, s, done);
            return launch_pipe4_kernel(m.kernel, b, s, done);
        }
        if (form == 2) return launch_split_kernels(m.has_model ? m.kernel : nullptr, a, s);
        return launch_stream_kernel(m.has_model ? m.kernel : nullptr, a, lds_bytes(m, a.mode == MODE_CHAIN ? a.n_frames : 0), s);
    }
    void release()
    {
        if (d_ctl) (void)hipFree(d_ctl);
        if (d_st) (void)hipFree(d_st);
        if (cur.d_nn) (void)hipFree(cur.d_nn);
        if (cur.d_wpack) (void)hipFree(cur.d_wpack);
        if (cur.d_wq4) (void)hipFree(cur.d_wq4);
        cur.d_wq4 = nullptr;
        if (cur.d_ring) (void)hipFree(cur.d_ring);
        if (cur.d_counters) (void)hipFree(cur.d_counters);
        if (cur.lp_owner) { lp_gate().release(device, cur.lp_owner); cur.lp_owner = nullptr; }
        if (h_lp_fault)                                      // (measurement builds / AIDAX_TUNE bit 4096: the stamps behind the fault word)
            if (const char* f = AIDAX_HOOK_ENV("AIDAX_LP_TRACE_FILE"))
                if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(h_lp_fault + 16, 4, 1536, fp); std::fclose(fp); }
        if (h_lp_fault) (void)hipHostFree(h_lp_fault);
        h_lp_fault = nullptr;
        for (const HostRange& r : host_ranges) (void)hipHostUnregister(r.base);
        host_ranges.clear();
        for (int k = 0; k < kPipeSets; ++k) {
            if (pipe.h_in[k]) (void)hipHostFree(pipe.h_in[k]);
            if (pipe.h_out[k]) (void)hipHostFree(pipe.h_out[k]);
            if (pipe.d_in[k]) (void)hipFree(pipe.d_in[k]);
            if (pipe.d_out[k]) (void)hipFree(pipe.d_out[k]);
            if (pipe.ev_up[k]) (void)hipEventDestroy(pipe.ev_up[k]);
            if (pipe.ev_pass[k]) (void)hipEventDestroy(pipe.ev_pass[k]);
            if (pipe.ev_down[k]) (void)hipEventDestroy(pipe.ev_down[k]);
        }
        if (pipe.q_up) (void)hipStreamDestroy(pipe.q_up);
        if (pipe.q_down) (void)hipStreamDestroy(pipe.q_down);
        pipe = Pipeline{};
        if (h_done) (void)hipHostFree(h_done);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
        for (int k = 0; k < kCtlRing; ++k) {
            if (ctl_ring[k]) (void)hipHostFree(ctl_ring[k]);
            if (ctl_ev[k]) (void)hipEventDestroy(ctl_ev[k]);
            ctl_ring[k] = nullptr; ctl_ev[k] = nullptr;
        }
        if (ev_x) (void)hipEventDestroy(ev_x);
        if (ev_adopt) (void)hipEventDestroy(ev_adopt);
        ev_adopt = nullptr;
        if (q) (void)hipStreamDestroy(q);
        if (wq) (void)hipStreamDestroy(wq);
        d_ctl = nullptr; d_st = nullptr; cur.d_nn = nullptr; cur.d_wpack = nullptr; d_in = nullptr; d_out = nullptr;
        cur.d_ring = nullptr; cur.d_counters = nullptr;
        h_in = nullptr; h_out = nullptr; ev_x = nullptr; q = nullptr; wq = nullptr;
    }
};

namespace {

void staged_release(aidax_staged* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->fenced) (void)hipEventSynchronize(s->fence);         // passes that still read the retired buffers
    if (s->slot.d_wpack) (void)hipFree(s->slot.d_wpack);
    if (s->slot.d_wq4) (void)hipFree(s->slot.d_wq4);
    if (s->slot.d_nn) (void)hipFree(s->slot.d_nn);
    if (s->slot.d_ring) (void)hipFree(s->slot.d_ring);
    if (s->slot.d_counters) (void)hipFree(s->slot.d_counters);
    if (s->slot.lp_owner) lp_gate().release(s->device, s->slot.lp_owner);
    if (s->d_pst) (void)hipFree(s->d_pst);
    if (s->fence) (void)hipEventDestroy(s->fence);
    delete s;
}

// loadModelFromPath's device half (rt-neural-generic.cpp:1034-1079) into buffers of its own. Worker thread:
// packs, allocates, uploads and warms up on the pool's worker stream and waits for it; touches nothing the
// audio side uses except for reading each stream's PARAM targets (as work() reads them at :822-825).
int prepare_impl(aidax_pool& p, const aidax_model* m, int start_mode, aidax_staged** out)
{
    HIP_TRY(hipSetDevice(p.device));
    std::unique_ptr<aidax_staged, void (*)(aidax_staged*)> sg(new aidax_staged(), staged_release);
    sg->device = p.device;
    sg->n_streams = p.n_streams;
    HIP_TRY(hipEventCreateWithFlags(&sg->fence, hipEventDisableTiming));
    if (!m) {                                             // unload: an empty slot to swap in
        *out = sg.release();
        return AIDAX_OK;
    }
    if (!model_supported(*m)) return fail(AIDAX_ERR_ARCH, "Unable to identify a known model architecture! (no kernel)");
    ModelSlot& ms = sg->slot;
    std::vector<float> wp;
    uint32_t state_floats = 0;
    if (is_conv_model(*m)) {
        ms.kind = ModelSlot::CONV;
        wp = pack_conv(*m, &ms.cdesc, &state_floats);
        ms.conv_mfma = p.force_form != 4 && convm_lds_bytes(ms.cdesc, p.ext_chunk()) <= 160 * 1024;
        // the stacks k_conv_ms admits run there (the contraction as bf16 term products: BASELINE cfg4 60 -> see profiles/r05_cfg4_*); its
        // per-layer histories live in the stream's state in ITS layout, so the choice holds for the life of the model in this pool
        // (warm-up, bare-model modes and passes all run the one kernel). AIDAX_CONV_MS=0 (test build): k_conv_mfma, the fp32 partner.
        const char* cms = AIDAX_HOOK_ENV("AIDAX_CONV_MS");
        ms.conv_ms = ms.conv_mfma && ms.cdesc.ms_ok && !(cms && cms[0] == '0');
        if (ms.conv_ms) state_floats = ms.cdesc.ms_state_floats;
        // ... and of those, the stacks with a compiled geometry (conv_st_shape) take fused blocks of 64 / 128 / 256 frames through k_conv_st,
        // the streaming form on the same state (AIDAX_CONV_ST=0, test build: k_conv_ms for every block)
        const char* cst = AIDAX_HOOK_ENV("AIDAX_CONV_ST");
        if (!ms.conv_ms || (cst && cst[0] == '0')) ms.cdesc.st_ok = 0;
        // chain passes inside the conv launch while every workgroup of the pool is resident at once (their serial
        // latency is then paid once per block; in a second round of workgroups it would be paid again, and the packed
        // k_chain launches around the kernel are cheaper). AIDAX_CONV_FUSED=1 / 0 forces the form.
        const char* fused = AIDAX_HOOK_ENV("AIDAX_CONV_FUSED");
        ms.conv_fused = ms.conv_mfma && (fused ? fused[0] != '0'
                                               : static_cast<int>(p.n_streams) <= (ms.conv_ms ? convs_resident_streams(p.device, ms.cdesc.st_ok - 1) : convm_resident_streams(ms.cdesc, p.ext_chunk(), p.device)));
        if (ms.conv_mfma && p.max_frames > 256) ms.conv_fused = true;      // long blocks go through in time slices: the one-launch form only
        if (!ms.conv_mfma && conv_lds_bytes(ms.cdesc, p.max_frames) > 160 * 1024)
            return fail(AIDAX_ERR_ARG, "conv model: pool max_frames too large for the LDS activation planes");
    } else if (m->n_rnn == 1 && find_kernel(m->cell, m->hidden) && p.quad_for_table_model(m->cell, m->hidden) &&
               chain_lds_bytes(p.max_frames) <= kChainLdsLimit && quad_lds_bytes(m->hidden, p.max_frames) <= 160 * 1024) {
        ms.kind = ModelSlot::QUAD;
        wp = pack_quad(*m, &ms.qdesc.bias_off, &ms.qdesc.dense_off);
        state_floats = static_cast<uint32_t>(m->cell == AIDAX_CELL_LSTM ? 2 * m->hidden : m->hidden);
    } else if (mfma_form_fits(*m) && chain_lds_bytes(p.max_frames) <= kChainLdsLimit &&
               (is_stack_model(*m) ? p.force_form != 4 : p.mfma_for_table_model(m->cell, m->hidden))) {
        ms.kind = ModelSlot::MFMA;
        wp = pack_mfma(*m, &ms.mdesc, &state_floats);
    } else if (is_stack_model(*m)) {
        ms.kind = ModelSlot::STACK;
        wp = pack_stack(*m, &ms.sdesc, &state_floats);
        if (stack_lds_bytes(ms.sdesc, p.max_frames) > 160 * 1024)
            return fail(AIDAX_ERR_ARG, "stacked model: pool max_frames too large for the LDS block buffers");
    } else {
        ms.kind = ModelSlot::TABLE;
        ms.kernel = find_kernel(m->cell, m->hidden);
        wp = pack_weights(*m);
        const int alt = (m->cell == AIDAX_CELL_LSTM && lstm_has_alt_pack(m->hidden)) ? lstm_pack_regs(m->hidden, false) * kWave : 0;   // a natural-order copy for k_nn
        if (static_cast<int>(wp.size()) != ms.kernel->pack_regs * kWave + m->hidden + 1 + alt) return fail(AIDAX_ERR_STATE, "weight pack size mismatch");
        state_floats = static_cast<uint32_t>(ms.kernel->state_floats);
    }
    ms.nn_stride = (state_floats + 3u) & ~3u;
    ms.cell = m->cell;
    ms.hidden = m->hidden;
    ms.input_size = m->input_size;
    ms.input_skip = m->input_skip;
    ms.in_gain = m->input_gain;
    ms.out_gain = m->output_gain;
    ms.model_sr = m->samplerate;
    ms.pipe_capacity = ms.kernel ? pipe_resident_streams(ms.kernel, p.max_frames, p.device) : 0;
    ms.split_pays = ms.kernel ? split_form_pays(ms.kernel, p.max_frames) : false;
    ms.has_model = true;

    HIP_TRY(hipMalloc(&ms.d_wpack, wp.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&ms.d_nn, static_cast<size_t>(p.n_streams) * ms.nn_stride * sizeof(float)));
    HIP_TRY(hipMalloc(&sg->d_pst, sizeof(StreamState) * p.n_streams));
    // Stacked model: one workgroup per (16 streams, layer), chained through a ring in global memory — where that adds
    // parallelism, i.e. while every (group, layer) workgroup gets a CU of its own (2048 streams for two layers on 256
    // CUs: 1.9x over one workgroup per group; beyond, where the CUs are full either way, a second round of workgroups
    // costs what the resident weights save: 4096 streams 2.49 against 2.45 ms). AIDAX_MFMA_LP=1 / 0 forces it on / off.
    const char* lp = AIDAX_HOOK_ENV("AIDAX_MFMA_LP");
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device));
    const size_t lp_groups = (p.n_streams + kMfmaStreams - 1) / kMfmaStreams;
    // (one-layer models of <= 48 units in the one-launch form: also in several rounds of workgroups — nothing waits for anything,
    // and LSTM-40 at 8192 streams measures 521 us against 583 us for k_mfma; profiles/r03_cfg3_forms.txt)
    const bool lp_rounds_ok = ms.kind == ModelSlot::MFMA && ms.mdesc.n_layers == 1 && ms.mdesc.hidden <= 48 && mfma_lp_fused_serves(ms.mdesc, p.max_frames);
    // (a one-layer model has no hand-over and nothing to wait for: no hold on the device's gate needed)
    const bool lp_chained = ms.mdesc.n_layers >= 2;
    // Round 4: k_mfma_ls is fast enough to pay in SEVERAL rounds too — a pass over more streams than one resident grid holds goes
    // out as one launch per range of streams (each a resident, cooperative grid; same ring, the counters are per group):
    // LSTM-96 x 2 at 4096 streams 2 x 0.71 ms against 2.48 ms on k_mfma, at 16 384 streams 8 x 0.71 against 9.42 ms.
    const char* sp_env = AIDAX_HOOK_ENV("AIDAX_LP_SPLIT");
    // (a lone layer on k_mfma_ls: AIDAX_LS1=1 / 0 forces it on / off)
    const char* ls1_env = AIDAX_HOOK_ENV("AIDAX_LS1");
    const bool ls1 = !lp_chained && (ls1_env ? ls1_env[0] != '0' : lone_split_pays(m->cell, m->hidden, p.n_streams, cus, p.max_frames));
    const bool ls_ok = ms.kind == ModelSlot::MFMA && (lp_chained || ls1) && mfma_ls_serves(ms.mdesc) && !(sp_env && sp_env[0] == '0') &&
                       mfma_ls_lds_bytes(ms.mdesc, p.max_frames) <= 160 * 1024;
    const char* rg_env = AIDAX_HOOK_ENV("AIDAX_LP_ROUND_GROUPS");      // (tests: ranges of this many stream groups, so that a small pool goes out in several)
    const size_t round_groups = rg_env && std::atoi(rg_env) > 0
                                    ? static_cast<size_t>(std::atoi(rg_env))
                                    : static_cast<size_t>(cus) / static_cast<size_t>(ms.mdesc.n_layers > 0 ? ms.mdesc.n_layers : 1) / 8 * 8;      // workgroup ids come in eights
    // ... where k_mfma_ls's time per resident grid beats k_mfma's rate on a full chip (profiles/r04_ls_rounds_ab.txt): 64 units and
    // more — LSTM-96 x 2 x1.65 / GRU-96 x 2 x1.64, 64 units x 2 x1.17..1.19, x 3 x1.10..1.16, GRU-80 x1.03; below 64 units k_mfma
    // wins by x1.4..1.65, and LSTM-80 (the four-wave geometry) pays up to two grids' worth of streams only
    const bool ranges_pay = rg_env || (ms.mdesc.hidden >= 64 && (ms.mdesc.hidden != 80 || ms.mdesc.L[0].cell != 0 || lp_groups <= static_cast<size_t>(cus)));
    const bool ls_rounds = ls_ok && ranges_pay && round_groups >= (rg_env ? 1u : 8u) && lp_groups > round_groups && !(lp && lp[0] == '1');
    // (the grid a chained launch asks for: workgroup ids come in eights, so the group count is rounded up — the padded workgroups return
    // at once, but a cooperative launch counts them: 85 groups x 3 layers are 264 workgroups, not 255)
    const size_t lp_blocks = (lp_groups + 7) / 8 * 8 * static_cast<size_t>(ms.mdesc.n_layers);
    const bool lp_pays = lp ? lp[0] != '0' : (lp_rounds_ok || ls_rounds || (ls_ok && !lp_chained) || lp_blocks <= static_cast<size_t>(cus));
    if (ms.kind == ModelSlot::MFMA && mfma_lp_serves(ms.mdesc) && lp_pays && !(lp_chained && p.lp_off.load()) &&
        mfma_lp_lds_bytes(ms.mdesc, p.max_frames) <= 160 * 1024 && (!lp_chained || lp_gate().acquire(p.device, &p, &p.lp_off))) {
        if (lp_chained) ms.lp_owner = &p;
        // stacked models whose split fragments fit the register file: the same hand-over with the contractions on the bf16 matrix
        // pipe (k_mfma_ls; AIDAX_LP_SPLIT=0: the fp32 kernel, =9: every term product instead of six — A/B runs)
        ms.lp_split = ls_ok ? (sp_env && sp_env[0] == '9' ? 9 : 6) : 0;
        ms.lp_round_streams = ls_rounds ? static_cast<uint32_t>(round_groups) * kMfmaStreams : 0u;
        const uint32_t ring_streams = ms.lp_round_streams ? ms.lp_round_streams : p.n_streams;
        const size_t ring_bytes = ms.lp_split ? mfma_ls_ring_bytes(ms.mdesc, ring_streams) : mfma_lp_ring_bytes(ms.mdesc, p.n_streams);
        HIP_TRY(hipMalloc(&ms.d_ring, ring_bytes));
        HIP_TRY(hipMalloc(&ms.d_counters, mfma_lp_counter_bytes(ms.mdesc, p.n_streams)));
        HIP_TRY(hipMemsetAsync(ms.d_counters, 0, mfma_lp_counter_bytes(ms.mdesc, p.n_streams), p.wq));
        const char* fu = AIDAX_HOOK_ENV("AIDAX_LP_FUSED");      // (=0: packed k_chain launches around the kernel, A/B runs)
        ms.lp_fused = !(fu && fu[0] == '0') &&
                      (ms.lp_split ? mfma_ls_fused_serves(ms.mdesc, p.max_frames) && mfma_ls_lds_bytes(ms.mdesc, p.max_frames, true) <= 160 * 1024
                                   : mfma_lp_fused_serves(ms.mdesc, p.max_frames) && mfma_lp_lds_bytes(ms.mdesc, p.max_frames, true) <= 160 * 1024);
    }
    if (ms.kind == ModelSlot::MFMA) {
        const char* gm = AIDAX_HOOK_ENV("AIDAX_GRU_GM");        // (=0: the four-rows-per-unit kernels; =f32: k_gru_gm with fp32 MFMAs — A/B runs)
        ms.gru_gm = gru_gm_serves(ms.mdesc) && !(gm && gm[0] == '0') && gru_gm_lds_bytes(ms.mdesc, p.max_frames) <= 160 * 1024;
        const char* np = AIDAX_HOOK_ENV("AIDAX_GS_PRODUCTS");   // (=9: every term product of the split operands instead of six)
        ms.lstm_gs = lstm_gs_serves(ms.mdesc) && lstm_gs_pays(m->cell, m->hidden, p.n_streams, cus) && lstm_gs_lds_bytes(ms.mdesc, p.max_frames) <= 160 * 1024
                         ? (np && np[0] == '9' ? 9 : 6) : 0;
        ms.gru_gs = gru_gs_serves(ms.mdesc) && !(gm && (gm[0] == 'f' || gm[0] == '0')) && gru_gs_lds_bytes(ms.mdesc, p.max_frames) <= 160 * 1024
                        ? (np && np[0] == '9' ? 9 : 6) : 0;
        if (ms.gru_gs) ms.gru_gm = true;                       // (80 units: the split kernel only — the flag says "one-launch gate-major form")
        else if (!gru_gm_serves(ms.mdesc)) ms.gru_gm = false;
    }
    HIP_TRY(hipMemcpyAsync(ms.d_wpack, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice, p.wq));
    std::vector<float> wq4;                                // (lives until the stream has been waited for below)
    if (ms.kind == ModelSlot::TABLE && q4_serves(m->cell, m->hidden, m->input_size) && p.force_form == 7 &&
        q4_lds_bytes(m->hidden, p.max_frames) <= 64 * 1024) {
        // AIDAX_KERNEL=q4 (A/B runs and tests only): LSTM-32 snapshot models, four streams per workgroup with the cell on the
        // matrix cores; the table kernels' record stays for warm-ups and the bare-model modes
        wq4 = pack_q4(*m);
        HIP_TRY(hipMalloc(&ms.d_wq4, wq4.size() * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(ms.d_wq4, wq4.data(), wq4.size() * sizeof(float), hipMemcpyHostToDevice, p.wq));
    }
    // fresh DynamicModel per stream: reset() + param smoothers around the targets the playing model holds now
    // (:822-825, :1035, :1053-1061) ...
    HIP_TRY(launch_stage_params(p.d_st, sg->d_pst, p.n_streams, p.wq));
    HIP_TRY(launch_reset_for_model(sg->d_pst, ms.d_nn, p.n_streams, ms.nn_stride, ms.p_den(), p.wq));
    if (start_mode == AIDAX_START_WARMUP) {               // ... and 2048 zeros through applyModel (:1077-1078)
        const uint32_t chunk = p.launch_chunk(ms, kWarmupFrames);
        for (uint32_t done = 0; done < kWarmupFrames; done += chunk) {
            LaunchArgs a = p.args(ms, sg->d_pst, nullptr, nullptr, std::min(chunk, kWarmupFrames - done), MODE_WARMUP);
            HIP_TRY(p.launch(ms, a, p.wq));
        }
    }
    HIP_TRY(hipStreamSynchronize(p.wq));                  // `wp` is pageable; the audio side must find the model complete
    *out = sg.release();
    return AIDAX_OK;
}

// work_response() (:859-893): swap, loading = false. Audio thread: host assignments plus one tiny kernel on the
// pool's stream; no allocation, no free, no wait.
int commit_impl(aidax_pool& p, aidax_staged* sg)
{
    HIP_TRY(hipSetDevice(p.device));
    p.enter_stream(p.q);
    if (sg->slot.has_model) HIP_TRY(launch_install_params(p.d_st, sg->d_pst, p.n_streams, p.q));
    HIP_TRY(hipEventRecord(sg->fence, p.q));              // the retired buffers are free once this has passed
    sg->fenced = true;
    std::swap(p.cur, sg->slot);
    const uint8_t loading = p.cur.has_model ? 0 : 1;      // :889 (unload: loading = true)
    for (uint32_t s = 0; s < p.n_streams; ++s) {
        p.loading[s] = loading;
        p.patch_model_fields(s);
    }
    p.mark_dirty(0, p.n_streams - 1);
    return AIDAX_OK;
}

}  // namespace

namespace aidax {

// aidax_pool_process_device over the first `n_active` streams only (rows of the others are neither read nor
// written, their state does not move): the hub launches what can be attached, not the pool's capacity.
int pool_process_prefix(aidax_pool* p, const float* d_in, float* d_out, uint32_t n_frames, void* hip_stream, uint32_t n_active)
{
    if (n_frames > p->max_frames) return fail(AIDAX_ERR_ARG, "n_frames exceeds the pool's max_frames");
    if (n_frames != 0 && (!d_in || !d_out)) return fail(AIDAX_ERR_ARG, "null buffer");
    if (n_active > p->n_streams) return fail(AIDAX_ERR_ARG, "n_active exceeds the pool's streams");
    if (n_active == 0) return AIDAX_OK;
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : p->q;
        p->enter_stream(s);
        p->flush_ctl(s);
        // a give-up of an earlier pass that nobody has collected yet: no further k_mfma_lp launch (its counters are out
        // of step); the word stays for whoever reports it (aidax_pool_sync, the hub)
        if (p->h_lp_fault && *static_cast<volatile uint32_t*>(p->h_lp_fault) != 0) p->lp_off.store(true, std::memory_order_relaxed);
        LaunchArgs a = p->args(p->cur, p->d_st, d_in, d_out, n_frames, MODE_CHAIN);
        a.n_streams = n_active;
        HIP_TRY(p->launch(p->cur, a, s));
        return AIDAX_OK;
    });
}

// Park / unpark one stream for the coming passes: while parked it behaves like a disabled plugin (raw copy, no state
// moves) whatever its own `enabled` control says. One flag flip in the host record — no filter redesign: the hub
// does this for every stream that is not part of the pass it launches.
int pool_park_stream(aidax_pool* p, uint32_t s, bool parked)
{
    if (s >= p->n_streams) return fail(AIDAX_ERR_ARG, "stream out of range");
    if ((p->forced_off[s] != 0) == parked) return AIDAX_OK;
    p->forced_off[s] = parked ? 1 : 0;
    StreamCtl& o = p->h_ctl[s];
    if (!parked && p->controls[s].enabled > 0.5f) o.flags |= CTL_ENABLED;
    else o.flags &= ~static_cast<uint32_t>(CTL_ENABLED);
    p->mark_dirty(s, s);
    return AIDAX_OK;
}

// Did a k_mfma_lp pass since the last call give up a hand-over (its output is wrong)? Clears the report and retires the
// kernel for this pool. The hub asks after a pass's event has passed.
bool pool_take_lp_fault(aidax_pool* p) { return p->take_lp_fault(); }
bool pool_chained_kernel_in_use(aidax_pool* p) { return p->cur.has_model && p->cur.kind == ModelSlot::MFMA && p->cur.mdesc.n_layers >= 2 && p->lp_in_use(p->cur); }
bool pool_lp_in_use(const aidax_pool* p) { return p->cur.has_model && p->lp_in_use(p->cur); }

}  // namespace aidax

extern "C" {

AIDAX_API int aidax_device_count(int* count)
{
    if (!count) return fail(AIDAX_ERR_ARG, "null count");
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(AIDAX_ERR_DEVICE, "no usable HIP device (this library has no CPU fallback)");
    *count = n;
    return AIDAX_OK;
}

AIDAX_API int aidax_pool_create(uint32_t n_streams, uint32_t max_frames, double host_samplerate,
                                int device_id, aidax_pool** out)
{
    if (!out) return fail(AIDAX_ERR_ARG, "null argument");
    *out = nullptr;
    if (n_streams == 0 || max_frames == 0 || max_frames > kMaxFrames || !(host_samplerate > 0))
        return fail(AIDAX_ERR_ARG, "n_streams/max_frames/samplerate out of range");
    return guarded([&]() -> int {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
            return fail(AIDAX_ERR_DEVICE, "no HIP device: the MI355X path has no CPU fallback");
        if (device_id < 0 || device_id >= n_dev) return fail(AIDAX_ERR_ARG, "device_id out of range");
        auto p = std::make_unique<aidax_pool>();
        p->device = device_id;
        p->n_streams = n_streams;
        p->max_frames = max_frames;
        p->host_sr = host_samplerate;
        p->gain_coef = exp_smoother_coef(static_cast<float>(host_samplerate), 0.1f);
        if (const char* f = AIDAX_HOOK_ENV("AIDAX_KERNEL"))
            p->force_form = std::strcmp(f, "wave") == 0 ? 1 : std::strcmp(f, "pipe") == 0 ? 2 : std::strcmp(f, "split") == 0 ? 3 : std::strcmp(f, "valu") == 0 ? 4 : std::strcmp(f, "mfma") == 0 ? 5 : std::strcmp(f, "quad") == 0 ? 6 : std::strcmp(f, "q4") == 0 ? 7 : 0;
        if (const char* t = AIDAX_HOOK_ENV("AIDAX_TUNE")) p->tune = std::atoi(t);
        try {
            HIP_TRY(hipSetDevice(device_id));
            HIP_TRY(hipDeviceGetAttribute(&p->cus, hipDeviceAttributeMultiprocessorCount, device_id));
            // the audio side's stream outranks the worker's: a pass must not queue behind a warm-up for its CUs
            int prio_least = 0, prio_greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            HIP_TRY(hipStreamCreateWithPriority(&p->q, hipStreamNonBlocking, prio_greatest));
            HIP_TRY(hipStreamCreateWithPriority(&p->wq, hipStreamNonBlocking, prio_least));
            { const char* sp = std::getenv("AIDAX_SPIN_WAIT"); p->spin_collect = !(sp && sp[0] == '0'); }
            { const char* kw = std::getenv("AIDAX_KERNEL_WORD"); p->kernel_word = !(kw && kw[0] == '0'); }
            p->keep_warm = keep_warm().acquire(device_id);
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_lp_fault), 8192, hipHostMallocDefault));     // the fault word + the time stamps of scratch/lp_trace.py, r05_modes.py
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->hd_lp_fault), p->h_lp_fault, 0));
            *p->h_lp_fault = 0;
            HIP_TRY(hipEventCreateWithFlags(&p->ev_x, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_adopt, hipEventDisableTiming));
            HIP_TRY(hipMalloc(&p->d_ctl, sizeof(StreamCtl) * n_streams));
            HIP_TRY(hipMalloc(&p->d_st, sizeof(StreamState) * n_streams));
            const size_t block_bytes = sizeof(float) * n_streams * static_cast<size_t>(max_frames);
            HIP_TRY(hipMalloc(&p->d_in, block_bytes));
            HIP_TRY(hipMalloc(&p->d_out, block_bytes));
            if (block_bytes <= kStagingLimit) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_in), block_bytes, hipHostMallocDefault));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_out), block_bytes, hipHostMallocDefault));
                const char* zc = std::getenv("AIDAX_ZEROCOPY");
                if (block_bytes <= kZeroCopyLimit && !(zc && zc[0] == '0')) {
                    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->hd_in), p->h_in, 0));
                    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->hd_out), p->h_out, 0));
                    p->zero_copy = true;
                    const char* sp = std::getenv("AIDAX_SPIN_WAIT");
                    int can = 0;
                    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device_id);
                    if (can && !(sp && sp[0] == '0')) {
                        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_done), 64, hipHostMallocDefault));
                        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->hd_done), p->h_done, 0));
                        *p->h_done = 0;
                        p->spin_wait = true;
                    }
                }
            }
            for (int k = 0; k < kCtlRing; ++k) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->ctl_ring[k]), sizeof(StreamCtl) * n_streams, hipHostMallocDefault));
                HIP_TRY(hipEventCreateWithFlags(&p->ctl_ev[k], hipEventDisableTiming));
            }
            p->controls.resize(n_streams);
            for (auto& c : p->controls) aidax_controls_default(&c);
            p->loading.assign(n_streams, 1);
            p->forced_off.assign(n_streams, 0);
            p->h_ctl.resize(n_streams);
            p->refresh_all();
            // instantiate(), rt-neural-generic.cpp:283-321: preGain target 1 cleared, masterGain target 0 cleared,
            // biquad z = 0, no model, loading = true
            HIP_TRY(launch_init_streams(p->d_st, n_streams, p->q));
            HIP_TRY(hipStreamSynchronize(p->q));
            p->last_stream = p->q;
        } catch (...) {
            p->release();
            throw;
        }
        *out = p.release();
        return AIDAX_OK;
    });
}

AIDAX_API void aidax_pool_destroy(aidax_pool* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->last_stream && p->last_stream != p->q) (void)hipStreamSynchronize(p->last_stream);
    if (p->q) (void)hipStreamSynchronize(p->q);
    if (p->wq) (void)hipStreamSynchronize(p->wq);
    if (p->pipe.q_up) (void)hipStreamSynchronize(p->pipe.q_up);
    if (p->pipe.q_down) (void)hipStreamSynchronize(p->pipe.q_down);
    p->release();
    if (p->keep_warm) keep_warm().release(p->device);
    delete p;
}

AIDAX_API uint32_t aidax_pool_streams(const aidax_pool* p) { return p ? p->n_streams : 0; }

AIDAX_API int aidax_pool_prepare_model(aidax_pool* p, const aidax_model* m, int start_mode, aidax_staged** out)
{
    if (!p || !out) return fail(AIDAX_ERR_ARG, "null argument");
    *out = nullptr;
    if (start_mode != AIDAX_START_WARMUP && start_mode != AIDAX_START_RESET) return fail(AIDAX_ERR_ARG, "bad start_mode");
    return guarded([&]() { return prepare_impl(*p, m, start_mode, out); });
}

AIDAX_API int aidax_pool_commit_model(aidax_pool* p, aidax_staged* staged)
{
    if (!p || !staged) return fail(AIDAX_ERR_ARG, "null argument");
    if (staged->fenced) return fail(AIDAX_ERR_STATE, "staged model was committed already");
    if (staged->device != p->device || staged->n_streams != p->n_streams) return fail(AIDAX_ERR_ARG, "staged model belongs to another pool");
    return guarded([&]() { return commit_impl(*p, staged); });
}

AIDAX_API void aidax_staged_free(aidax_staged* staged) { staged_release(staged); }

AIDAX_API int aidax_pool_set_model(aidax_pool* p, const aidax_model* m, int start_mode)
{
    aidax_staged* sg = nullptr;
    int rc = aidax_pool_prepare_model(p, m, start_mode, &sg);
    if (rc != AIDAX_OK) return rc;
    rc = aidax_pool_commit_model(p, sg);
    aidax_staged_free(sg);                                  // what the swap retired (or, on failure, the unused model)
    return rc;
}

AIDAX_API int aidax_pool_reset_stream(aidax_pool* p, uint32_t stream, int start_mode)
{
    return aidax::pool_reset_stream_inherit(p, stream, start_mode, nullptr);
}

}  // extern "C"

namespace aidax {

// aidax_pool_reset_stream for a stream that continues another plugin instance's life: the fresh DynamicModel is built
// around `p_targets` (the PARAM targets of the model that instance plays now, :822-825) instead of the 0 / 0 of a first load.
int pool_reset_stream_inherit(aidax_pool* p, uint32_t stream, int start_mode, const float* p_targets)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (stream >= p->n_streams) return fail(AIDAX_ERR_ARG, "stream out of range");
    if (start_mode != AIDAX_START_WARMUP && start_mode != AIDAX_START_RESET) return fail(AIDAX_ERR_ARG, "bad start_mode");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        p->enter_stream(p->q);
        const ModelSlot& m = p->cur;
        HIP_TRY(launch_init_streams(p->d_st + stream, 1, p->q));      // instantiate(), :283-321
        if (p_targets) HIP_TRY(launch_set_param_targets(p->d_st + stream, p_targets[0], p_targets[1], p->q));
        if (m.has_model) {
            // a fresh DynamicModel for this stream only: the launch arguments view the pool as one stream
            HIP_TRY(launch_reset_for_model(p->d_st + stream, m.d_nn + static_cast<size_t>(stream) * m.nn_stride, 1,
                                           m.nn_stride, m.p_den(), p->q));
            if (start_mode == AIDAX_START_WARMUP) {
                const uint32_t chunk = p->launch_chunk(m, kWarmupFrames);
                for (uint32_t done = 0; done < kWarmupFrames; done += chunk) {
                    LaunchArgs a = p->args(m, p->d_st, nullptr, nullptr, std::min(chunk, kWarmupFrames - done), MODE_WARMUP);
                    a.ctl += stream; a.st += stream; a.nn += static_cast<size_t>(stream) * m.nn_stride;
                    a.n_streams = 1;
                    HIP_TRY(p->launch(m, a, p->q));
                }
            }
        }
        p->loading[stream] = m.has_model ? 0 : 1;
        p->refresh_ctl(stream);
        return AIDAX_OK;
    });
}

// Stream `ds` of `dst` takes over the plugin-owned DSP members (biquad states, gain smoothers) of stream `ss` of `src`,
// as they are once everything `src` has issued so far has run: an event on src's latest stream, a wait on dst's own,
// one tiny kernel. Nothing waits on the host: this is work_response() of an instance that moves between hubs. The
// caller serialises access to both pools (the hubs' locks).
int pool_adopt_stream_dsp(aidax_pool* dst, uint32_t ds, aidax_pool* src, uint32_t ss)
{
    if (!dst || !src || ds >= dst->n_streams || ss >= src->n_streams) return fail(AIDAX_ERR_ARG, "adopt: stream out of range");
    if (dst->device != src->device) return fail(AIDAX_ERR_ARG, "adopt: pools on different devices");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(dst->device));
        hipStream_t from = src->last_stream ? src->last_stream : src->q;
        HIP_TRY(hipEventRecord(src->ev_adopt, from));
        dst->enter_stream(dst->q);
        HIP_TRY(hipStreamWaitEvent(dst->q, src->ev_adopt, 0));
        HIP_TRY(launch_adopt_dsp(dst->d_st + ds, src->d_st + ss, dst->q));
        if (src != dst) {                                   // src must not overwrite the record before the copy has read it:
            HIP_TRY(hipEventRecord(dst->ev_adopt, dst->q));  // its next operation waits for the copy (the stream is parked or
            src->enter_stream(src->q);                       // detached by then, but a later occupant of the seat is reset on src->q)
            HIP_TRY(hipStreamWaitEvent(src->q, dst->ev_adopt, 0));
        }
        return AIDAX_OK;
    });
}

// One stream's record as it is once the pool's issued work has run (waits: worker / test side only).
int pool_read_stream_state(aidax_pool* p, uint32_t stream, StreamState* out)
{
    if (!p || !out || stream >= p->n_streams) return fail(AIDAX_ERR_ARG, "stream out of range");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        if (p->last_stream && p->last_stream != p->q) HIP_TRY(hipStreamSynchronize(p->last_stream));
        HIP_TRY(hipStreamSynchronize(p->q));
        HIP_TRY(hipMemcpy(out, p->d_st + stream, sizeof(StreamState), hipMemcpyDeviceToHost));
        return AIDAX_OK;
    });
}

bool pool_has_model(const aidax_pool* p) { return p->cur.has_model; }

// The same read without a wait inside: the copy (into pinned memory of the caller) and `done` are put behind what the
// pool has issued so far; the caller waits for `done` wherever it may.
int pool_peek_stream_state(aidax_pool* p, uint32_t stream, StreamState* pinned_out, void* done_event)
{
    if (!p || !pinned_out || stream >= p->n_streams) return fail(AIDAX_ERR_ARG, "stream out of range");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        p->enter_stream(p->q);
        HIP_TRY(hipMemcpyAsync(pinned_out, p->d_st + stream, sizeof(StreamState), hipMemcpyDeviceToHost, p->q));
        HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(done_event), p->q));
        return AIDAX_OK;
    });
}

}  // namespace aidax

extern "C" {

AIDAX_API int aidax_pool_export_stream_dsp(aidax_pool* p, uint32_t stream, aidax_stream_dsp* out)
{
    if (!p || !out) return fail(AIDAX_ERR_ARG, "null argument");
    StreamState st{};
    const int rc = aidax::pool_read_stream_state(p, stream, &st);
    if (rc != AIDAX_OK) return rc;
    std::memcpy(out->z, st.z, sizeof(out->z));
    out->pre_mem = st.pre_mem; out->master_mem = st.master_mem;
    out->pre_target = st.pre_tgt; out->master_target = st.master_tgt;
    out->param_target[0] = p->cur.has_model ? st.p_tgt[0] : 0.f;           // no model: work() passes 0 / 0 (:815-816)
    out->param_target[1] = p->cur.has_model ? st.p_tgt[1] : 0.f;
    return AIDAX_OK;
}

AIDAX_API int aidax_pool_import_stream_dsp(aidax_pool* p, uint32_t stream, const aidax_stream_dsp* in)
{
    if (!p || !in) return fail(AIDAX_ERR_ARG, "null argument");
    if (stream >= p->n_streams) return fail(AIDAX_ERR_ARG, "stream out of range");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        StreamState st{};
        const int rc = aidax::pool_read_stream_state(p, stream, &st);     // waits for the pool's passes: not an audio-thread call
        if (rc != AIDAX_OK) return rc;
        std::memcpy(st.z, in->z, sizeof(st.z));
        st.pre_mem = in->pre_mem; st.master_mem = in->master_mem;
        st.pre_tgt = in->pre_target; st.master_tgt = in->master_target;
        st.pending &= ~static_cast<uint32_t>(PEND_ACTIVATE);
        HIP_TRY(hipMemcpy(p->d_st + stream, &st, sizeof(StreamState), hipMemcpyHostToDevice));
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_set_loading(aidax_pool* p, int32_t stream, int loading)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (stream != AIDAX_ALL_STREAMS && (stream < 0 || static_cast<uint32_t>(stream) >= p->n_streams))
        return fail(AIDAX_ERR_ARG, "stream out of range");
    const uint32_t lo = stream == AIDAX_ALL_STREAMS ? 0u : static_cast<uint32_t>(stream);
    const uint32_t hi = stream == AIDAX_ALL_STREAMS ? p->n_streams - 1 : static_cast<uint32_t>(stream);
    for (uint32_t s = lo; s <= hi; ++s) {
        p->loading[s] = loading ? 1 : 0;
        p->patch_model_fields(s);
    }
    p->mark_dirty(lo, hi);
    return AIDAX_OK;
}

AIDAX_API int aidax_pool_set_controls(aidax_pool* p, int32_t stream, const aidax_controls* c)
{
    if (!p || !c) return fail(AIDAX_ERR_ARG, "null argument");
    if (stream != AIDAX_ALL_STREAMS && (stream < 0 || static_cast<uint32_t>(stream) >= p->n_streams))
        return fail(AIDAX_ERR_ARG, "stream out of range");
    if (stream == AIDAX_ALL_STREAMS) {
        for (auto& dst : p->controls) dst = *c;
        p->refresh_all();
    } else {
        p->controls[stream] = *c;
        p->refresh_ctl(static_cast<uint32_t>(stream));
    }
    return AIDAX_OK;
}

AIDAX_API int aidax_pool_activate(aidax_pool* p, int32_t stream)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (stream != AIDAX_ALL_STREAMS && (stream < 0 || static_cast<uint32_t>(stream) >= p->n_streams))
        return fail(AIDAX_ERR_ARG, "stream out of range");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        p->enter_stream(p->q);
        // paramFirstRun is only re-armed when a model exists (rt-neural-generic.cpp:344-351)
        const uint32_t bits = PEND_ACTIVATE | (p->cur.has_model ? PEND_PARAM_FIRST : 0u);
        HIP_TRY(launch_set_pending(p->d_st, p->n_streams, stream, bits, p->q));
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_process_device(aidax_pool* p, const float* d_in, float* d_out, uint32_t n_frames, void* hip_stream)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    return pool_process_prefix(p, d_in, d_out, n_frames, hip_stream, p->n_streams);
}

AIDAX_API int aidax_pool_process(aidax_pool* p, const float* in, float* out, uint32_t n_frames)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (n_frames > p->max_frames) return fail(AIDAX_ERR_ARG, "n_frames exceeds the pool's max_frames");
    if (n_frames != 0 && (!in || !out)) return fail(AIDAX_ERR_ARG, "null buffer");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        const size_t bytes = sizeof(float) * p->n_streams * static_cast<size_t>(n_frames);
        int rc;
        if (!p->h_in) {                                     // blocks beyond the pinned staging: pageable copies
            if (bytes) HIP_TRY(hipMemcpyAsync(p->d_in, in, bytes, hipMemcpyHostToDevice, p->q));
            rc = aidax_pool_process_device(p, p->d_in, p->d_out, n_frames, p->q);
            if (rc != AIDAX_OK) return rc;
            if (bytes) HIP_TRY(hipMemcpyAsync(out, p->d_out, bytes, hipMemcpyDeviceToHost, p->q));
            HIP_TRY(hipStreamSynchronize(p->q));
            if (p->take_lp_fault()) {
                if (bytes) std::memset(out, 0, bytes);
                return fail(AIDAX_ERR_DEVICE, "k_mfma_lp: a layer hand-over timed out (this block is silence; the pool falls back to k_mfma)");
            }
            return AIDAX_OK;
        }
        if (bytes) std::memcpy(p->h_in, in, bytes);
        if (p->zero_copy) {
            // small blocks (the one-instance plugin): the pass reads its input straight from pinned host memory;
            // a one-launch pass also writes its output there, a multi-launch pass works in place on d_out
            const bool direct = p->single_launch(p->cur);
            p->pass_word_taken = false;
            if (direct && p->spin_wait && p->kernel_word) { p->pass_word = p->hd_done; p->pass_seq = p->done_seq + 1; }
            rc = aidax_pool_process_device(p, p->hd_in, direct ? p->hd_out : p->d_out, n_frames, p->q);
            p->pass_word = nullptr;
            if (rc != AIDAX_OK) return rc;
            if (!direct && bytes) HIP_TRY(hipMemcpyAsync(p->h_out, p->d_out, bytes, hipMemcpyDeviceToHost, p->q));
        } else {
            if (bytes) HIP_TRY(hipMemcpyAsync(p->d_in, p->h_in, bytes, hipMemcpyHostToDevice, p->q));
            rc = aidax_pool_process_device(p, p->d_in, p->d_out, n_frames, p->q);
            if (rc != AIDAX_OK) return rc;
            if (bytes) HIP_TRY(hipMemcpyAsync(p->h_out, p->d_out, bytes, hipMemcpyDeviceToHost, p->q));
        }
        uint32_t seq = 0;
        if (p->spin_wait && p->pass_word_taken) {
            seq = ++p->done_seq;                            // (the pass writes it itself)
            p->pass_word_taken = false;
        } else if (p->spin_wait) {
            seq = ++p->done_seq;
            if (hipStreamWriteValue32(p->q, p->hd_done, seq, 0) != hipSuccess) {      // a runtime without stream memory operations
                (void)hipGetLastError();
                p->spin_wait = false;
            }
        }
        if (p->spin_wait) {
            // poll the completion word for up to ~2 ms (a block that takes longer is not a real-time block), then wait the usual way
            volatile uint32_t* w = p->h_done;
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t polls = 0;
            while (*w != seq) {
                if ((++polls & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    HIP_TRY(hipStreamSynchronize(p->q));
                    break;
                }
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            HIP_TRY(hipStreamSynchronize(p->q));
        }
        if (p->take_lp_fault()) {                           // the pass is wrong: silence, and k_mfma from the next one on
            if (bytes) std::memset(out, 0, bytes);
            return fail(AIDAX_ERR_DEVICE, "k_mfma_lp: a layer hand-over timed out (this block is silence; the pool falls back to k_mfma)");
        }
        if (bytes) std::memcpy(out, p->h_out, bytes);
        return AIDAX_OK;
    });
}

// ---- the pipelined host-buffer path. One caller thread (the audio side); at most two blocks between submit and collect.
static int pool_submit_impl(aidax_pool* p, const float* in, float* out, uint32_t n_frames)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (n_frames == 0 || n_frames > p->max_frames) return fail(AIDAX_ERR_ARG, "n_frames out of range");
    if (!in) return fail(AIDAX_ERR_ARG, "null buffer");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        auto& pl = p->pipe;
        if (!pl.ready) {                                    // first call: the second staging set (not a real-time call)
            const size_t cap = sizeof(float) * p->n_streams * static_cast<size_t>(p->max_frames);
            HIP_TRY(hipStreamCreateWithFlags(&pl.q_up, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&pl.q_down, hipStreamNonBlocking));
            for (int k = 0; k < aidax_pool::kPipeSets; ++k) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pl.h_in[k]), cap, hipHostMallocDefault));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pl.h_out[k]), cap, hipHostMallocDefault));
                HIP_TRY(hipMalloc(&pl.d_in[k], cap));
                HIP_TRY(hipMalloc(&pl.d_out[k], cap));
                HIP_TRY(hipEventCreateWithFlags(&pl.ev_up[k], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&pl.ev_pass[k], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&pl.ev_down[k], hipEventDisableTiming));
            }
            pl.ready = true;
        }
        if (pl.submitted - pl.collected >= static_cast<uint64_t>(aidax_pool::kPipeSets)) return fail(AIDAX_ERR_STATE, "three blocks are in flight: collect one first");
        const int s = static_cast<int>(pl.submitted % aidax_pool::kPipeSets);
        const size_t bytes = sizeof(float) * p->n_streams * static_cast<size_t>(n_frames);
        // set s was last used by block k-3, which has been collected: its pass and both its copies are complete
        const float* up_from = in;                          // a registered caller buffer is uploaded as it lies
        if (!p->host_registered(in, bytes)) {
            std::memcpy(pl.h_in[s], in, bytes);
            up_from = pl.h_in[s];
        }
        HIP_TRY(hipMemcpyAsync(pl.d_in[s], up_from, bytes, hipMemcpyHostToDevice, pl.q_up));
        HIP_TRY(hipEventRecord(pl.ev_up[s], pl.q_up));
        p->enter_stream(p->q);
        HIP_TRY(hipStreamWaitEvent(p->q, pl.ev_up[s], 0));
        // (a launch that can carry the "pass done" event on its dispatch packet saves the marker packet behind the pass: 78 -> 72.5 us per
        // cfg2 block, profiles/r06_host_pipeline.txt)
        p->pass_done = pl.ev_pass[s];
        p->pass_done_taken = false;
        const int rc = pool_process_prefix(p, pl.d_in[s], pl.d_out[s], n_frames, p->q, p->n_streams);
        p->pass_done = nullptr;
        if (rc != AIDAX_OK) return rc;
        if (!p->pass_done_taken) HIP_TRY(hipEventRecord(pl.ev_pass[s], p->q));
        HIP_TRY(hipStreamWaitEvent(pl.q_down, pl.ev_pass[s], 0));
        pl.direct_out[s] = (out && p->host_registered(out, bytes)) ? out : nullptr;
        HIP_TRY(hipMemcpyAsync(pl.direct_out[s] ? pl.direct_out[s] : pl.h_out[s], pl.d_out[s], bytes, hipMemcpyDeviceToHost, pl.q_down));
        HIP_TRY(hipEventRecord(pl.ev_down[s], pl.q_down));
        pl.frames[s] = n_frames;
        ++pl.submitted;
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_submit(aidax_pool* p, const float* in, uint32_t n_frames) { return pool_submit_impl(p, in, nullptr, n_frames); }

AIDAX_API int aidax_pool_submit_to(aidax_pool* p, const float* in, float* out, uint32_t n_frames)
{
    if (!out) return fail(AIDAX_ERR_ARG, "null buffer");
    return pool_submit_impl(p, in, out, n_frames);
}

AIDAX_API int aidax_pool_register_host(aidax_pool* p, void* base, size_t bytes)
{
    if (!p || !base || bytes == 0) return fail(AIDAX_ERR_ARG, "null range");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        for (const auto& r : p->host_ranges)
            if (r.base == static_cast<char*>(base)) return fail(AIDAX_ERR_STATE, "range is registered already");
        HIP_TRY(hipHostRegister(base, bytes, hipHostRegisterDefault));
        p->host_ranges.push_back({ static_cast<char*>(base), bytes });
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_unregister_host(aidax_pool* p, void* base)
{
    if (!p || !base) return fail(AIDAX_ERR_ARG, "null range");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        for (size_t i = 0; i < p->host_ranges.size(); ++i)
            if (p->host_ranges[i].base == static_cast<char*>(base)) {
                // copies to or from the range that are still in flight finish first
                if (p->pipe.q_up) HIP_TRY(hipStreamSynchronize(p->pipe.q_up));
                if (p->pipe.q_down) HIP_TRY(hipStreamSynchronize(p->pipe.q_down));
                HIP_TRY(hipHostUnregister(base));
                p->host_ranges.erase(p->host_ranges.begin() + static_cast<long>(i));
                return AIDAX_OK;
            }
        return fail(AIDAX_ERR_ARG, "range was not registered");
    });
}

AIDAX_API int aidax_pool_collect(aidax_pool* p, float* out, uint32_t n_frames)
{
    if (!p || !out) return fail(AIDAX_ERR_ARG, "null argument");
    return guarded([&]() -> int {
        auto& pl = p->pipe;
        if (!pl.ready || pl.collected == pl.submitted) return fail(AIDAX_ERR_STATE, "nothing was submitted");
        const int s = static_cast<int>(pl.collected % aidax_pool::kPipeSets);
        if (pl.frames[s] != n_frames) return fail(AIDAX_ERR_ARG, "n_frames differs from the submitted block's");
        if (pl.direct_out[s] && pl.direct_out[s] != out) return fail(AIDAX_ERR_ARG, "collect: the block was submitted with another destination");
        HIP_TRY(hipSetDevice(p->device));
        if (hipEventQuery(pl.ev_down[s]) != hipSuccess) {
            // like aidax_pool_process: poll (no interrupt, no wake-up on the way back) for up to ~2 ms, then wait the usual way
            bool done = false;
            if (p->spin_collect) {
                const auto t0 = std::chrono::steady_clock::now();
                uint32_t polls = 0;
                while (!(done = hipEventQuery(pl.ev_down[s]) == hipSuccess)) {
                    if ((++polls & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
                (void)hipGetLastError();                        // (hipErrorNotReady of the polls)
            }
            if (!done) HIP_TRY(hipEventSynchronize(pl.ev_down[s]));
        }
        ++pl.collected;
        const size_t bytes = sizeof(float) * p->n_streams * static_cast<size_t>(n_frames);
        if (p->take_lp_fault()) {
            std::memset(out, 0, bytes);
            return fail(AIDAX_ERR_DEVICE, "k_mfma_lp: a layer hand-over timed out (this block is silence; the pool falls back to k_mfma)");
        }
        if (!pl.direct_out[s]) std::memcpy(out, pl.h_out[s], bytes);
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_sync(aidax_pool* p)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        if (p->last_stream && p->last_stream != p->q) HIP_TRY(hipStreamSynchronize(p->last_stream));
        HIP_TRY(hipStreamSynchronize(p->q));
        if (p->pipe.q_down) HIP_TRY(hipStreamSynchronize(p->pipe.q_down));
        // k_mfma_lp: did a layer hand-over of a pass since the last report give up waiting?
        if (p->take_lp_fault()) return fail(AIDAX_ERR_DEVICE, "k_mfma_lp: a layer hand-over timed out (a pass since the last sync is invalid; the pool falls back to k_mfma)");
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_read_state(aidax_pool* p, uint32_t stream, int layer, float* h, float* c, uint32_t cap)
{
    if (!p || !h) return fail(AIDAX_ERR_ARG, "null argument");
    const ModelSlot& m = p->cur;
    if (!m.has_model || stream >= p->n_streams || m.kind == ModelSlot::CONV) return fail(AIDAX_ERR_STATE, "no such state");
    // TABLE and QUAD pools keep one layer as h[H] | c[H] (pack_quad shares the table kernels' state layout)
    const int n_layers = m.kind == ModelSlot::MFMA ? m.mdesc.n_layers : m.kind == ModelSlot::STACK ? m.sdesc.n_layers : 1;
    if (layer < 0 || layer >= n_layers) return fail(AIDAX_ERR_STATE, "no such layer");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        if (p->last_stream && p->last_stream != p->q) HIP_TRY(hipStreamSynchronize(p->last_stream));
        HIP_TRY(hipStreamSynchronize(p->q));
        uint32_t H = static_cast<uint32_t>(m.hidden), off = 0;
        bool lstm = false;
        if (m.kind == ModelSlot::STACK) {
            H = static_cast<uint32_t>(m.sdesc.L[layer].hidden); off = m.sdesc.L[layer].state_off; lstm = m.sdesc.L[layer].cell == 0;
        } else if (m.kind == ModelSlot::MFMA) {
            H = static_cast<uint32_t>(m.mdesc.hidden_true); off = m.mdesc.L[layer].state_off; lstm = m.mdesc.L[layer].cell == 0;
        } else {
            lstm = m.cell == AIDAX_CELL_LSTM;
        }
        const uint32_t n = H < cap ? H : cap;
        const float* base = m.d_nn + static_cast<size_t>(stream) * m.nn_stride + off;
        HIP_TRY(hipMemcpy(h, base, n * sizeof(float), hipMemcpyDeviceToHost));
        if (c && lstm) HIP_TRY(hipMemcpy(c, base + H, n * sizeof(float), hipMemcpyDeviceToHost));
        return static_cast<int>(H);
    });
}

AIDAX_API int aidax_many_streams_form(int cell, int hidden, uint32_t n_streams, int compute_units)
{
    return static_cast<int>(many_streams_form(cell, hidden, n_streams, compute_units, 256));
}
AIDAX_API int aidax_many_streams_form_at(int cell, int hidden, uint32_t n_streams, int compute_units, uint32_t max_frames)
{
    return static_cast<int>(many_streams_form(cell, hidden, n_streams, compute_units, max_frames));
}

AIDAX_API int aidax_model_conv_form(const aidax_model* m)
{
    if (!m || !is_conv_model(*m)) return 0;
    ConvDesc d;
    uint32_t state_floats = 0;
    (void)pack_conv(*m, &d, &state_floats);
    if (convm_lds_bytes(d, 256) > 160 * 1024) return 1;
    return d.st_ok ? 4 : d.ms_ok ? 3 : 2;
}

AIDAX_API const char* aidax_pool_kernel_name(const aidax_pool* p)
{
    if (!(p && p->cur.has_model)) return "k_nomodel";
    const ModelSlot& m = p->cur;
    if (m.kind == ModelSlot::STACK) return "k_stack";
    if (m.kind == ModelSlot::MFMA && m.lstm_gs) return "k_lstm_gs";
    if (m.kind == ModelSlot::MFMA) return m.gru_gm ? (m.gru_gs ? "k_gru_gs" : "k_gru_gm") : !p->lp_in_use(m) ? "k_chain+k_mfma" : m.lp_split ? (m.mdesc.n_layers == 1 ? (m.lp_fused ? "k_mfma_ls1" : "k_chain+k_mfma_ls1") : m.lp_fused ? "k_mfma_ls" : "k_chain+k_mfma_ls") : m.lp_fused ? "k_mfma_lp" : "k_chain+k_mfma_lp";
    if (m.kind == ModelSlot::QUAD) return "k_chain+k_quad";
    if (m.kind == ModelSlot::CONV && m.conv_ms && m.conv_fused && m.cdesc.st_ok && p->max_frames >= 64) return "k_conv_st";      // (what a block of 64 / 128 / 256 frames runs; every other length: k_conv_ms)
    if (m.kind == ModelSlot::CONV) return m.conv_ms ? (m.conv_fused ? "k_conv_ms" : "k_chain+k_conv_ms") : m.conv_fused ? "k_conv_mfma" : m.conv_mfma ? "k_chain+k_conv_mfma" : "k_conv";
    const int form = p->chain_form(m);
    // (form 1: what a block of the pool's full length runs with the controls as they stand — k_*_pipe4 where it serves, k_*_pipe otherwise)
    if (form == 1 && p->pipe4_serves(m, p->max_frames, m.input_size)) return m.kernel->name_pipe4;
    return form == 3 ? "k_lstm_q4<32>" : form == 1 ? m.kernel->name_pipe : form == 2 ? m.kernel->name_split : m.kernel->name;
}

AIDAX_API int aidax_model_forward(const aidax_model* m, int device_id, const float* X, float* y, uint32_t n, int unit_gains)
{
    if (!m || !X || !y) return fail(AIDAX_ERR_ARG, "null argument");
    aidax_pool* p = nullptr;
    int rc = aidax_pool_create(1, 4, 48000.0, device_id, &p);
    if (rc != AIDAX_OK) return rc;
    rc = aidax_pool_set_model(p, m, AIDAX_START_RESET);
    if (rc == AIDAX_OK) {
        rc = guarded([&]() -> int {
            float* d_x = nullptr;
            float* d_y = nullptr;
            const size_t xb = sizeof(float) * static_cast<size_t>(n) * m->input_size;
            HIP_TRY(hipMalloc(&d_x, xb ? xb : 4));
            HIP_TRY(hipMalloc(&d_y, sizeof(float) * (n ? n : 1)));
            HIP_TRY(hipMemcpyAsync(d_x, X, xb, hipMemcpyHostToDevice, p->q));
            hipError_t le = hipSuccess;
            const uint32_t chunk = p->launch_chunk(p->cur, n ? n : 1);
            for (uint32_t done = 0; done < n && le == hipSuccess; done += chunk) {
                LaunchArgs a = p->args(p->cur, p->d_st, d_x + static_cast<size_t>(done) * m->input_size, d_y + done,
                                       std::min(chunk, n - done), MODE_NN_ONLY);
                if (unit_gains) { a.in_gain = 1.f; a.out_gain = 1.f; }
                le = p->launch(p->cur, a, p->q);
            }
            if (le == hipSuccess) {
                (void)hipMemcpyAsync(y, d_y, sizeof(float) * n, hipMemcpyDeviceToHost, p->q);
            }
            const hipError_t se = hipStreamSynchronize(p->q);
            (void)hipFree(d_x);
            (void)hipFree(d_y);
            HIP_TRY(le);
            HIP_TRY(se);
            return AIDAX_OK;
        });
    }
    aidax_pool_destroy(p);
    return rc;
}

AIDAX_API int aidax_model_self_test(const aidax_model* m, int device_id, int32_t* n_errors, float* max_error, float* out_opt)
{
    if (!m || !n_errors || !max_error) return fail(AIDAX_ERR_ARG, "null argument");
    const size_t n = m->golden_in.size();
    if (n == 0) return fail(AIDAX_ERR_STATE, "model carries no input_batch/output_batch");
    // params forced to 0, gains forced to 1 (rt-neural-generic.cpp:903-915)
    std::vector<float> X(n * static_cast<size_t>(m->input_size), 0.f);
    for (size_t t = 0; t < n; ++t) X[t * m->input_size] = m->golden_in[t];
    std::vector<float> y(n);
    const int rc = aidax_model_forward(m, device_id, X.data(), y.data(), static_cast<uint32_t>(n), 1);
    if (rc != AIDAX_OK) return rc;
    int32_t errs = 0;
    float worst = 0.f;
    for (size_t t = 0; t < n; ++t) {
        const float e = std::fabs(y[t] - m->golden_out[t]);
        if (e > worst) worst = e;
        if (static_cast<double>(e) > 1.0e-5) ++errs;          // TEST_MODEL_THR, rt-neural-generic.h:182
    }
    *n_errors = errs;
    *max_error = worst;
    if (out_opt) std::memcpy(out_opt, y.data(), n * sizeof(float));
    return AIDAX_OK;
}

}  // extern "C"
