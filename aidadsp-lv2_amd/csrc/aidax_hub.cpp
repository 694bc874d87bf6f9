// aidax_hub.cpp — host-side stream aggregator: many plugin instances of one process, one pool pass per
// audio period (include/aidax.h, "hub"). Pipelined by one period so that hosts that call their instances
// one after another never wait on each other; see the header for the contract.
//
// Who does what:
//   run() of an instance   stages its input row into pinned memory and counts itself in (a few hundred
//                          nanoseconds under the hub's mutex), then — outside the mutex — waits for the event of
//                          the PREVIOUS period's pass (normally long complete) and copies its output row out.
//                          It launches nothing unless it has to close a period itself (re-entry before the
//                          launcher got to it, change of block size).
//   the launcher thread    launches the pass of a period (H2D of the attached rows, the pool pass, D2H, event)
//                          as soon as every attached instance has submitted, or when the period's deadline
//                          passes — so one instance that stalls or stops calling cannot hold the others' audio.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "aidax_internal.h"

using namespace aidax;
using Clock = std::chrono::steady_clock;

// Staging buffers in rotation. One pass per period needs two (collect into one while the other's output is read); the
// deadline may close a period in pieces (a host whose submissions span more than the deadline), and then an
// instance's output of the last period sits in the buffer of the piece IT was part of, not in the most recent one —
// so every slot remembers the pass that carried its last block, and a buffer is reused only kHubBuffers passes later
// (its output stays readable for kHubBuffers - 2 further passes: not while it is being collected into again).
constexpr int kHubBuffers = 4;

struct aidax_hub {
    aidax_pool* pool = nullptr;
    uint32_t cap = 0, max_frames = 0;
    int device = 0;
    double host_sr = 48000.0;
    std::mutex mu;
    std::condition_variable cv;
    std::thread launcher;
    bool stop = false, flush_requested = false;
    int64_t deadline_us = -1;                    // < 0: half the period; 0: no deadline
    Clock::time_point deadline{};
    std::vector<uint8_t> attached, submitted, forced_off;
    std::vector<uint64_t> last_pass;             // per slot: id of the pass that carried its latest block (0: none)
    std::vector<aidax_controls> ctl;
    uint32_t n_attached = 0, n_submitted = 0, period_frames = 0, out_frames[kHubBuffers] = {};
    uint64_t pass_id[kHubBuffers] = {};          // which pass a buffer holds
    uint32_t hi_slot = 0;                        // rows [0, hi_slot) can be attached: what a pass moves and launches
    uint32_t latency = 0;
    uint64_t launches = 0, deadline_launches = 0;
    uint64_t last_chained = 0;                   // the latest pass that went out on a chained kernel (k_mfma_lp / k_mfma_ls): later ones cannot have given up
    uint64_t clean_upto = 0;                     // passes up to this id are known to carry no k_mfma_lp give-up
    uint64_t faults_unmapped = 0;                // give-up reports that mapped to no chained pass (nothing to silence; counted so that none vanishes)
    uint64_t bad_from = 1, bad_upto = 0;         // passes in [bad_from, bad_upto] may: their rows are delivered as silence
    int last_error = AIDAX_OK;
    float* h_in[kHubBuffers] = {};               // pinned staging, rows packed at the period's block length
    float* h_out[kHubBuffers] = {};
    float* d_in = nullptr;
    float* d_out = nullptr;
    hipStream_t q = nullptr;
    hipEvent_t done[kHubBuffers] = {};
    bool guard[kHubBuffers] = {};                // the buffer's previous pass may still be reading its input staging: whoever
                                                 // stages into it next waits for done[b] first — outside the lock (aidax_hub_run)
    StreamState* h_peek = nullptr;               // pinned: one seat's record on its way to a worker thread (attach_successor)
    hipEvent_t peek_ev = nullptr;
    std::mutex peek_mu;                          // one reader at a time
    int cur = 1;                                 // the buffer being collected into: always (id of the next pass) % kHubBuffers

    ~aidax_hub()
    {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv.notify_all();
        if (launcher.joinable()) launcher.join();
        if (q) (void)hipStreamSynchronize(q);
        if (pool) aidax_pool_destroy(pool);      // before the stream it last ran on goes away
        for (int i = 0; i < kHubBuffers; ++i) {
            if (h_in[i]) (void)hipHostFree(h_in[i]);
            if (h_out[i]) (void)hipHostFree(h_out[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
        if (h_peek) (void)hipHostFree(h_peek);
        if (peek_ev) (void)hipEventDestroy(peek_ev);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (q) (void)hipStreamDestroy(q);
    }
};

namespace {

bool ok(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HUB_TRY(x) do { if (!ok((x), #x)) return AIDAX_ERR_DEVICE; } while (0)

// A slot that is detached, or did not take part in the period being launched, is parked for that pass (a parked
// stream is a raw copy: its state does not move); its controls stay what the instance set.
int push_controls(aidax_hub& h, uint32_t slot, bool off)
{
    h.forced_off[slot] = off ? 1 : 0;
    return pool_park_stream(h.pool, slot, off);
}

int launch_period(aidax_hub& h);

// launch the period that is being collected (h.mu held): asynchronous, nothing in here waits for the GPU unless the
// host has run kHubBuffers - 1 passes ahead of it. A period
// that cannot be launched is dropped (its instances read silence for it and the error is reported by their next
// run()): the launcher must never find the same failing period waiting for it again.
int flush_locked(aidax_hub& h)
{
    h.flush_requested = false;
    if (h.n_submitted == 0) return AIDAX_OK;
    const int rc = launch_period(h);
    if (rc != AIDAX_OK) {
        std::fill(h.submitted.begin(), h.submitted.end(), 0);
        h.n_submitted = 0;
    }
    return rc;
}

int launch_period(aidax_hub& h)
{
    const uint32_t n = h.period_frames;
    const uint32_t rows = h.hi_slot;
    for (uint32_t s = 0; s < rows; ++s) {
        const bool off = !h.attached[s] || !h.submitted[s];
        if (off != (h.forced_off[s] != 0)) {
            const int rc = push_controls(h, s, off);
            if (rc != AIDAX_OK) return rc;
        }
    }
    const int b = h.cur;
    const size_t bytes = sizeof(float) * static_cast<size_t>(rows) * n;
    HUB_TRY(hipSetDevice(h.device));
    if (bytes != 0) HUB_TRY(hipMemcpyAsync(h.d_in, h.h_in[b], bytes, hipMemcpyHostToDevice, h.q));
    const int rc = pool_process_prefix(h.pool, h.d_in, h.d_out, n, h.q, rows);
    if (rc != AIDAX_OK) return rc;
    if (pool_chained_kernel_in_use(h.pool)) h.last_chained = h.launches + 1;      // (asked after the launch: a pool that has just fallen back answers no)
    if (bytes != 0) HUB_TRY(hipMemcpyAsync(h.h_out[b], h.d_out, bytes, hipMemcpyDeviceToHost, h.q));
    HUB_TRY(hipEventRecord(h.done[b], h.q));
    ++h.launches;
    h.out_frames[b] = n;
    h.pass_id[b] = h.launches;
    for (uint32_t s = 0; s < rows; ++s)
        if (h.submitted[s]) h.last_pass[s] = h.launches;
    std::fill(h.submitted.begin(), h.submitted.end(), 0);
    h.n_submitted = 0;
    h.latency = n;
    h.cur = static_cast<int>((h.launches + 1) % kHubBuffers);
    // The buffer that is collected into next was the staging of the pass kHubBuffers back: its rows must have left for
    // the device before the instances overwrite them. A host that closes several short periods in a row (block-size
    // changes, skipped instances) can run that far ahead of the GPU — found by tests/soak_hub.py as outputs computed from
    // a later block's input. Normally that pass is long complete and this is one event query; otherwise the buffer is
    // marked and the first run() that stages into it waits for the event, with the lock released.
    h.guard[h.cur] = h.pass_id[h.cur] != 0 && hipEventQuery(h.done[h.cur]) != hipSuccess;
    return AIDAX_OK;
}

void launcher_main(aidax_hub* h)
{
    std::unique_lock<std::mutex> lk(h->mu);
    while (!h->stop) {
        const bool timed = h->n_submitted != 0 && h->deadline_us != 0;
        if (h->flush_requested || (timed && Clock::now() >= h->deadline)) {
            if (!h->flush_requested) ++h->deadline_launches;
            const int rc = flush_locked(*h);
            if (rc != AIDAX_OK) h->last_error = rc;
            continue;
        }
        if (timed) h->cv.wait_until(lk, h->deadline);
        else h->cv.wait(lk);
    }
}

// counters read under the mutex: the launcher thread writes them
template <class T, class F>
T hub_read(const aidax_hub* h, F&& f)
{
    if (!h) return T{};
    aidax_hub* m = const_cast<aidax_hub*>(h);
    std::lock_guard<std::mutex> g(m->mu);
    return f(*m);
}

}  // namespace

extern "C" {

AIDAX_API int aidax_hub_create(uint32_t max_instances, uint32_t max_frames, double host_samplerate, int device_id, aidax_hub** out)
{
    if (!out) return fail(AIDAX_ERR_ARG, "null argument");
    *out = nullptr;
    aidax_pool* pool = nullptr;
    const int rc = aidax_pool_create(max_instances, max_frames, host_samplerate, device_id, &pool);
    if (rc != AIDAX_OK) return rc;
    aidax_hub* h = new aidax_hub();
    h->pool = pool;
    h->cap = max_instances;
    h->max_frames = max_frames;
    h->device = device_id;
    h->host_sr = host_samplerate;
    h->attached.assign(max_instances, 0);
    h->submitted.assign(max_instances, 0);
    h->forced_off.assign(max_instances, 0);
    h->last_pass.assign(max_instances, 0);
    h->ctl.resize(max_instances);
    for (auto& c : h->ctl) aidax_controls_default(&c);
    const size_t bytes = sizeof(float) * static_cast<size_t>(max_instances) * max_frames;
    bool good = ok(hipSetDevice(device_id), "hipSetDevice") &&
                ok(hipStreamCreateWithFlags(&h->q, hipStreamNonBlocking), "hipStreamCreate") &&
                ok(hipMalloc(&h->d_in, bytes), "hipMalloc") && ok(hipMalloc(&h->d_out, bytes), "hipMalloc");
    for (int i = 0; i < kHubBuffers && good; ++i)
        good = ok(hipHostMalloc(reinterpret_cast<void**>(&h->h_in[i]), bytes, hipHostMallocDefault), "hipHostMalloc") &&
               ok(hipHostMalloc(reinterpret_cast<void**>(&h->h_out[i]), bytes, hipHostMallocDefault), "hipHostMalloc") &&
               ok(hipEventCreateWithFlags(&h->done[i], hipEventDisableTiming), "hipEventCreate");
    good = good && ok(hipHostMalloc(reinterpret_cast<void**>(&h->h_peek), sizeof(StreamState), hipHostMallocDefault), "hipHostMalloc") &&
           ok(hipEventCreateWithFlags(&h->peek_ev, hipEventDisableTiming), "hipEventCreate");
    if (!good) { delete h; return AIDAX_ERR_DEVICE; }
    // nobody is attached yet: every stream rests disabled
    for (uint32_t s = 0; s < max_instances; ++s)
        if (push_controls(*h, s, true) != AIDAX_OK) { delete h; return AIDAX_ERR_DEVICE; }
    h->launcher = std::thread(launcher_main, h);
    *out = h;
    return AIDAX_OK;
}

AIDAX_API void aidax_hub_destroy(aidax_hub* h) { delete h; }

AIDAX_API int aidax_hub_set_deadline_us(aidax_hub* h, int64_t microseconds)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->deadline_us = microseconds;
    }
    h->cv.notify_all();
    return AIDAX_OK;
}

AIDAX_API int aidax_hub_set_model(aidax_hub* h, const aidax_model* m, int start_mode)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    // the worker half needs no lock: it builds into buffers of its own on the pool's worker stream
    aidax_staged* sg = nullptr;
    int rc = aidax_pool_prepare_model(h->pool, m, start_mode, &sg);
    if (rc != AIDAX_OK) return rc;
    {
        std::lock_guard<std::mutex> g(h->mu);
        rc = flush_locked(*h);                              // blocks submitted so far were played by the old model
        if (rc == AIDAX_OK) rc = aidax_pool_commit_model(h->pool, sg);
    }
    aidax_staged_free(sg);
    return rc;
}

namespace {

int attach_locked(aidax_hub* h, int32_t* slot, const float* p_targets)
{
    for (uint32_t s = 0; s < h->cap; ++s) {
        if (h->attached[s]) continue;
        // a fresh instance in this slot: a handful of asynchronous launches on the pool's stream, ordered
        // against the passes by the pool's stream edges — nothing here waits for the GPU
        int rc = pool_reset_stream_inherit(h->pool, s, AIDAX_START_WARMUP, p_targets);
        if (rc != AIDAX_OK) return rc;
        rc = aidax_pool_activate(h->pool, static_cast<int32_t>(s));
        if (rc != AIDAX_OK) return rc;
        aidax_controls_default(&h->ctl[s]);
        rc = aidax_pool_set_controls(h->pool, static_cast<int32_t>(s), &h->ctl[s]);     // not the previous occupant's
        if (rc != AIDAX_OK) return rc;
        h->attached[s] = 1;
        h->last_pass[s] = 0;
        ++h->n_attached;
        if (s + 1 > h->hi_slot) h->hi_slot = s + 1;
        *slot = static_cast<int32_t>(s);
        return AIDAX_OK;
    }
    return fail(AIDAX_ERR_STATE, "hub is full");
}

}  // namespace

AIDAX_API int aidax_hub_attach(aidax_hub* h, int32_t* slot)
{
    if (!h || !slot) return fail(AIDAX_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    return attach_locked(h, slot, nullptr);
}

// Worker thread of an instance that plays on (prev, prev_slot) and has loaded another model file: a seat in `h` whose
// fresh DynamicModel is built around the PARAM targets its playing model holds (work(), :822-825, :1053-1061 — they
// are also the conditioning inputs of the 2048-zero warm-up). The predecessor's pending block is launched first, the
// read waits for it outside every lock.
AIDAX_API int aidax_hub_attach_successor(aidax_hub* h, aidax_hub* prev, int32_t prev_slot, int32_t* slot)
{
    if (!h || !slot) return fail(AIDAX_ERR_ARG, "null argument");
    float targets[2] = { 0.f, 0.f };                         // no predecessor / no model there: work() passes 0 / 0
    if (prev) {
        std::lock_guard<std::mutex> one(prev->peek_mu);
        bool reading = false;
        {
            std::lock_guard<std::mutex> g(prev->mu);
            if (prev_slot >= 0 && static_cast<uint32_t>(prev_slot) < prev->cap && prev->attached[prev_slot] && pool_has_model(prev->pool)) {
                if (prev->submitted[prev_slot]) {
                    const int rc = flush_locked(*prev);
                    if (rc != AIDAX_OK) return rc;
                }
                const int rc = pool_peek_stream_state(prev->pool, static_cast<uint32_t>(prev_slot), prev->h_peek, prev->peek_ev);
                if (rc != AIDAX_OK) return rc;
                reading = true;
            }
        }
        if (reading) {
            HUB_TRY(hipEventSynchronize(prev->peek_ev));
            targets[0] = prev->h_peek->p_tgt[0];
            targets[1] = prev->h_peek->p_tgt[1];
        }
    }
    std::lock_guard<std::mutex> g(h->mu);
    return attach_locked(h, slot, targets);
}

// Audio thread, work_response() of that instance (:859-893): from its next block on it plays on (h, slot). The
// plugin's own DSP members — the seven biquads' memories and both gain smoothers (rt-neural-generic.h:311-317), which
// the reference keeps across a model swap (:868-875) — move with it: the predecessor's pending block is launched, and a
// device-side copy ordered behind it fills the new seat's record. No wait, no allocation.
AIDAX_API int aidax_hub_adopt(aidax_hub* h, int32_t slot, aidax_hub* prev, int32_t prev_slot)
{
    if (!h || !prev) return fail(AIDAX_ERR_ARG, "null argument");
    auto body = [&]() -> int {
        if (slot < 0 || static_cast<uint32_t>(slot) >= h->cap || !h->attached[slot]) return fail(AIDAX_ERR_ARG, "slot not attached");
        if (prev_slot < 0 || static_cast<uint32_t>(prev_slot) >= prev->cap || !prev->attached[prev_slot]) return fail(AIDAX_ERR_ARG, "predecessor not attached");
        if (prev->submitted[prev_slot]) {                    // its last block belongs to the stream the new seat continues
            const int rc = flush_locked(*prev);
            if (rc != AIDAX_OK) return rc;
        }
        return pool_adopt_stream_dsp(h->pool, static_cast<uint32_t>(slot), prev->pool, static_cast<uint32_t>(prev_slot));
    };
    if (h == prev) {
        std::lock_guard<std::mutex> g(h->mu);
        return body();
    }
    std::scoped_lock g(h->mu, prev->mu);
    return body();
}

AIDAX_API int aidax_hub_detach(aidax_hub* h, int32_t slot)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    std::lock_guard<std::mutex> g(h->mu);
    if (slot < 0 || static_cast<uint32_t>(slot) >= h->cap || !h->attached[slot]) return fail(AIDAX_ERR_ARG, "slot not attached");
    h->attached[slot] = 0;
    --h->n_attached;
    if (h->submitted[slot]) { h->submitted[slot] = 0; --h->n_submitted; }
    if (h->n_submitted != 0 && h->n_submitted == h->n_attached) return flush_locked(*h);
    return AIDAX_OK;
}

AIDAX_API int aidax_hub_set_controls(aidax_hub* h, int32_t slot, const aidax_controls* c)
{
    if (!h || !c) return fail(AIDAX_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (slot < 0 || static_cast<uint32_t>(slot) >= h->cap || !h->attached[slot]) return fail(AIDAX_ERR_ARG, "slot not attached");
    if (std::memcmp(&h->ctl[slot], c, sizeof(*c)) == 0) return AIDAX_OK;
    // the instance's last block may still wait for its pass (somebody else has not submitted yet): that block was
    // played under the OLD controls, so its period is closed before the new ones go in
    if (h->submitted[slot]) {
        const int rc = flush_locked(*h);
        if (rc != AIDAX_OK) return rc;
    }
    h->ctl[slot] = *c;
    return aidax_pool_set_controls(h->pool, slot, c);      // the pool keeps the slot parked if it is
}

AIDAX_API int aidax_hub_set_loading(aidax_hub* h, int32_t slot, int loading)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    std::lock_guard<std::mutex> g(h->mu);
    if (slot < 0 || static_cast<uint32_t>(slot) >= h->cap || !h->attached[slot]) return fail(AIDAX_ERR_ARG, "slot not attached");
    if (h->submitted[slot]) {                               // as for the controls: what was submitted plays under what was in force
        const int rc = flush_locked(*h);
        if (rc != AIDAX_OK) return rc;
    }
    return aidax_pool_set_loading(h->pool, slot, loading);
}

AIDAX_API int aidax_hub_activate(aidax_hub* h, int32_t slot)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    std::lock_guard<std::mutex> g(h->mu);
    if (slot < 0 || static_cast<uint32_t>(slot) >= h->cap || !h->attached[slot]) return fail(AIDAX_ERR_ARG, "slot not attached");
    if (h->submitted[slot]) {
        const int rc = flush_locked(*h);
        if (rc != AIDAX_OK) return rc;
    }
    return aidax_pool_activate(h->pool, slot);
}

AIDAX_API int aidax_hub_run(aidax_hub* h, int32_t slot, const float* in, float* out, uint32_t n_frames)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    if (n_frames > h->max_frames) return fail(AIDAX_ERR_ARG, "n_frames exceeds the hub's max_frames");
    if (n_frames != 0 && (!in || !out)) return fail(AIDAX_ERR_ARG, "null buffer");
    const float* prev_row = nullptr;
    hipEvent_t prev_done = nullptr;
    uint64_t prev_pass = 0;
    int prev_buf = 0;
    bool wake = false;
    {
        std::unique_lock<std::mutex> g(h->mu);
        for (;;) {
            if (slot < 0 || static_cast<uint32_t>(slot) >= h->cap || !h->attached[slot]) return fail(AIDAX_ERR_ARG, "slot not attached");
            if (h->last_error != AIDAX_OK) {                 // a pass the launcher could not issue
                const int rc = h->last_error;
                h->last_error = AIDAX_OK;
                return fail(rc, "hub: a pool pass failed to launch");
            }
            // this instance is back before the period was closed (somebody was skipped and the launcher has not
            // got to it yet), or the host changed the block size: close the period here
            if (h->submitted[slot] || (h->n_submitted != 0 && n_frames != h->period_frames)) {
                const int rc = flush_locked(*h);
                if (rc != AIDAX_OK) return rc;
            }
            // the staging buffer of the period being collected: if the pass that last used it (kHubBuffers back) has not
            // read its rows yet — a host far ahead of the GPU — wait for it WITHOUT the lock (the others keep running),
            // then look at everything again
            const int gb = h->cur;
            if (!h->guard[gb]) break;
            if (hipEventQuery(h->done[gb]) == hipSuccess) { h->guard[gb] = false; break; }
            hipEvent_t ev = h->done[gb];
            g.unlock();
            const hipError_t we = hipEventSynchronize(ev);
            g.lock();
            if (we != hipSuccess) return fail(AIDAX_ERR_DEVICE, "hub: waiting for a pass failed");
        }
        if (h->n_submitted == 0) {                           // first of a new period: block length and deadline
            h->period_frames = n_frames;
            const int64_t us = h->deadline_us < 0 ? static_cast<int64_t>(0.5e6 * n_frames / h->host_sr) : h->deadline_us;
            h->deadline = Clock::now() + std::chrono::microseconds(us);
            wake = h->deadline_us != 0;
        }
        const int b = h->cur;
        if (n_frames != 0) std::memcpy(h->h_in[b] + static_cast<size_t>(slot) * n_frames, in, sizeof(float) * n_frames);
        h->submitted[slot] = 1;
        ++h->n_submitted;
        // the pass that carried this instance's previous block, if its buffer has not been reused since
        const uint64_t lp = h->last_pass[slot];
        const int pb = static_cast<int>(lp % kHubBuffers);
        // (not from the buffer being collected into: the pass that closes this period will write its results there)
        if (n_frames != 0 && lp != 0 && pb != h->cur && h->pass_id[pb] == lp && h->out_frames[pb] == n_frames) {
            prev_row = h->h_out[pb] + static_cast<size_t>(slot) * n_frames;
            prev_done = h->done[pb];
            prev_pass = lp;
            prev_buf = pb;
        }
        if (h->n_submitted == h->n_attached) { h->flush_requested = true; wake = true; }
    }
    if (wake) h->cv.notify_one();
    // the previous period's output for this instance; its pass was launched a period ago
    if (n_frames != 0) {
        bool delivered = false, lp_fault = false;
        if (prev_row) {
            if (hipEventQuery(prev_done) != hipSuccess) HUB_TRY(hipEventSynchronize(prev_done));
            // The row is copied under the mutex, after checking that the buffer still holds that pass: while this thread
            // waited, further periods may have closed (deadline, another instance's block size, set_controls, detach),
            // and the pass kHubBuffers later is copied into this very buffer — a row read without the check could be
            // half of each. (A thousand floats under the lock: a fraction of a microsecond.)
            std::lock_guard<std::mutex> g(h->mu);
            if (pool_take_lp_fault(h->pool)) {              // some pass in (clean_upto, launches] gave up a layer hand-over ...
                h->bad_from = h->clean_upto + 1;
                h->bad_upto = std::min(h->launches, h->last_chained);      // ... and only one that ran on a chained kernel can have
                if (h->bad_upto < h->bad_from) ++h->faults_unmapped;       // (a report that maps to no pass still counts: aidax_hub_faults_unmapped)
            }
            lp_fault = prev_pass >= h->bad_from && prev_pass <= h->bad_upto;
            if (!lp_fault && prev_pass > h->clean_upto) h->clean_upto = prev_pass;     // its event has passed and nothing was reported
            if (!lp_fault && h->pass_id[prev_buf] == prev_pass) {
                std::memcpy(out, prev_row, sizeof(float) * n_frames);
                delivered = true;
            }
        }
        if (!prev_row) {
            // no previous block to hand back (first block on the seat, a changed block length): a pending give-up is taken
            // here all the same, so that it marks the passes it belongs to and not a later, clean one
            std::lock_guard<std::mutex> g(h->mu);
            if (pool_take_lp_fault(h->pool)) {
                h->bad_from = h->clean_upto + 1;
                h->bad_upto = std::min(h->launches, h->last_chained);
                if (h->bad_upto < h->bad_from) ++h->faults_unmapped;
            }
        }
        if (!delivered) std::memset(out, 0, sizeof(float) * n_frames);
        if (lp_fault) return fail(AIDAX_ERR_DEVICE, "hub: k_mfma_lp gave up a layer hand-over in the pass of this block (silence; the pool falls back to k_mfma)");
    }
    return AIDAX_OK;
}

AIDAX_API int aidax_hub_flush(aidax_hub* h)
{
    if (!h) return fail(AIDAX_ERR_ARG, "null hub");
    std::lock_guard<std::mutex> g(h->mu);
    return flush_locked(*h);
}

AIDAX_API uint32_t aidax_hub_latency_frames(const aidax_hub* h) { return hub_read<uint32_t>(h, [](aidax_hub& x) { return x.latency; }); }
AIDAX_API uint32_t aidax_hub_max_frames(const aidax_hub* h) { return h ? h->max_frames : 0; }
AIDAX_API uint32_t aidax_hub_attached(const aidax_hub* h) { return hub_read<uint32_t>(h, [](aidax_hub& x) { return x.n_attached; }); }
AIDAX_API uint64_t aidax_hub_launches(const aidax_hub* h) { return hub_read<uint64_t>(h, [](aidax_hub& x) { return x.launches; }); }
AIDAX_API uint64_t aidax_hub_deadline_launches(const aidax_hub* h) { return hub_read<uint64_t>(h, [](aidax_hub& x) { return x.deadline_launches; }); }
AIDAX_API uint64_t aidax_hub_faults_unmapped(const aidax_hub* h) { return hub_read<uint64_t>(h, [](aidax_hub& x) { return x.faults_unmapped; }); }

}  // extern "C"
