// aidax_pool.cpp — the stream pool behind the C ABI: device residency, control
// latching, model swap and the process launches. Host logic only; the kernels
// are in aidax_kernels.hip. No CPU fallback: any HIP failure is AIDAX_ERR_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
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
enum ManyForm { MANY_NONE = 0, MANY_QUAD = 1, MANY_MFMA = 2 };
ManyForm many_streams_form(int cell, int hidden, uint32_t n)
{
    const bool lstm = cell == AIDAX_CELL_LSTM;
    if (hidden <= 32) return n >= 4096 ? MANY_QUAD : MANY_NONE;
    if (hidden == 40) return n >= 8192 ? MANY_MFMA : n >= (lstm ? 512u : 4096u) ? MANY_QUAD : MANY_NONE;
    if (hidden == 64 && !lstm) return n >= 16384 ? MANY_MFMA : n <= 1024 ? MANY_QUAD : MANY_NONE;
    // LSTM-64, LSTM-80, GRU-80: their one-wave kernels hold 250-500 weight registers; k_quad is 1.35-1.75x
    // faster already at 64 streams, k_mfma takes over from 4096
    return n >= 4096 ? MANY_MFMA : MANY_QUAD;
}

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

}  // namespace
}  // namespace aidax

struct aidax_pool {
    int device = 0;
    uint32_t n_streams = 0, max_frames = 0;
    double host_sr = 48000.0;
    float gain_coef = 0.f;
    hipStream_t q = nullptr;

    StreamCtl* d_ctl = nullptr;
    StreamState* d_st = nullptr;
    float* d_nn = nullptr;
    float* d_wpack = nullptr;
    float* d_in = nullptr;           // staging for the host-buffer entry point
    float* d_out = nullptr;

    std::vector<aidax_controls> controls;
    std::vector<uint8_t> loading;
    std::vector<StreamCtl> h_ctl;
    bool ctl_dirty = true;

    // model in use (copy of what the kernels need)
    bool has_model = false;
    enum Kind { TABLE = 0, STACK = 1, CONV = 2, MFMA = 3, QUAD = 4 } kind = TABLE;
    QuadDesc qdesc{};
    int cell = 0;
    StackDesc sdesc{};
    MfmaDesc mdesc{};
    ConvDesc cdesc{};
    bool conv_mfma = false;          // conv stacks on the matrix cores (blocks of <= 256 frames), else k_conv
    const KernelEntry* kernel = nullptr;
    int input_size = 1, input_skip = 0, hidden = 0;
    float in_gain = 1.f, out_gain = 1.f, model_sr = 48000.f;
    uint32_t nn_stride = 0;
    int pipe_capacity = 0;           // streams the 3-wave pipeline keeps resident at once (0 = never use it)
    bool split_pays = false;         // the lean recurrent kernel keeps the occupancy of the one-wave kernel
    int force_form = 0;              // AIDAX_KERNEL=wave|pipe|split|valu overrides the heuristic (A/B testing)
    // Form of a MODE_CHAIN pass of a TABLE pool: 0 one wave per stream, 1 three-wave pipeline (all streams resident at
    // once: latency-bound regime), 2 split launches with packed chains (many streams: issue-bound regime)
    int chain_form() const
    {
        if (kind != TABLE && has_model) return 0;
        if (chain_lds_bytes(max_frames) > kChainLdsLimit) return 0;      // packed chains keep 8 blocks in LDS
        if (force_form == 1) return 0;
        if (force_form == 2) return (has_model && kernel) ? 1 : 0;
        if (force_form == 3) return 2;
        if (use_pipe()) return 1;
        if (n_streams < 64) return 0;
        return (!has_model || split_pays) ? 2 : 0;
    }
    // Models of the reference's table on the matrix-core kernels (many_streams_form); AIDAX_KERNEL=quad|mfma force one.
    // four streams per workgroup on mfma_f32_4x4x1 (k_quad): the many-streams form of the table models
    bool quad_for_table_model(int cell_kind, int hidden_units) const
    {
        if (force_form == 6) return true;
        return force_form == 0 && many_streams_form(cell_kind, hidden_units, n_streams) == MANY_QUAD;
    }
    bool mfma_for_table_model(int cell_kind, int hidden_units) const
    {
        if (force_form == 5) return true;
        return force_form == 0 && many_streams_form(cell_kind, hidden_units, n_streams) == MANY_MFMA;
    }
    bool use_pipe() const
    {
        if (!has_model || !kernel || kind != TABLE) return false;
        if (force_form == 1 || force_form == 3) return false;
        if (force_form == 2) return true;
        return static_cast<int>(n_streams) <= pipe_capacity;
    }

    size_t lds_bytes(uint32_t n_frames) const
    {
        return (static_cast<size_t>((n_frames + 3) & ~3u) + static_cast<size_t>(hidden > 0 ? hidden : 4)) * sizeof(float);
    }
    float p_den() const { return 0.1f * model_sr; }      // LinearValueSmoother tau * sampleRate (:1053-1054)

    void refresh_ctl(uint32_t s)
    {
        build_stream_ctl(controls[s], host_sr, has_model, loading[s] != 0, gain_coef, p_den(), &h_ctl[s]);
        ctl_dirty = true;
    }
    void refresh_all()
    {
        // identical controls are the common case: design once, then copy
        for (uint32_t s = 0; s < n_streams; ++s) {
            if (s > 0 && std::memcmp(&controls[s], &controls[s - 1], sizeof(aidax_controls)) == 0 &&
                loading[s] == loading[s - 1]) {
                h_ctl[s] = h_ctl[s - 1];
            } else {
                refresh_ctl(s);
            }
        }
        ctl_dirty = true;
    }
    void flush_ctl(hipStream_t s)
    {
        if (!ctl_dirty) return;
        HIP_TRY(hipMemcpyAsync(d_ctl, h_ctl.data(), sizeof(StreamCtl) * n_streams, hipMemcpyHostToDevice, s));
        ctl_dirty = false;
    }
    LaunchArgs args(const float* in, float* out, uint32_t n_frames, int mode) const
    {
        LaunchArgs a{};
        a.ctl = d_ctl; a.st = d_st; a.nn = d_nn; a.wpack = d_wpack;
        a.in = in; a.out = out;
        a.n_streams = n_streams; a.n_frames = n_frames; a.nn_stride = nn_stride;
        a.mode = mode; a.input_size = input_size; a.input_skip = input_skip;
        a.in_gain = in_gain; a.out_gain = out_gain;
        return a;
    }
    // frames one launch of the extension kernels may carry (their LDS planes grow with n)
    uint32_t ext_chunk() const { return max_frames < 256 ? max_frames : 256; }
    uint32_t launch_chunk(uint32_t n) const { return (kind == STACK || kind == CONV) ? std::min(ext_chunk(), n) : n; }
    hipError_t launch(const LaunchArgs& a, hipStream_t s) const
    {
        if (has_model && kind == MFMA) {
            // split form around the matrix-core kernel: packed chains in -> out, applyModel in place, packed chains
            if (a.mode != MODE_CHAIN) return launch_mfma_kernel(a, mdesc, s);
            hipError_t e = launch_chain_pass(true, a, s);
            if (e == hipSuccess && a.n_frames != 0) e = launch_mfma_kernel(a, mdesc, s);
            if (e == hipSuccess) e = launch_chain_pass(false, a, s);
            return e;
        }
        if (has_model && kind == QUAD) {
            if (a.mode != MODE_CHAIN) return launch_quad_kernel(cell, hidden, a, qdesc, s);
            hipError_t e = launch_chain_pass(true, a, s);
            if (e == hipSuccess && a.n_frames != 0) e = launch_quad_kernel(cell, hidden, a, qdesc, s);
            if (e == hipSuccess) e = launch_chain_pass(false, a, s);
            return e;
        }
        if (has_model && kind == STACK) return launch_stack_kernel(a, sdesc, s);
        if (has_model && kind == CONV && conv_mfma) {
            if (a.mode != MODE_CHAIN) return launch_conv_mfma_kernel(a, cdesc, s);
            hipError_t e = launch_chain_pass(true, a, s);
            if (e == hipSuccess && a.n_frames != 0) e = launch_conv_mfma_kernel(a, cdesc, s);
            if (e == hipSuccess) e = launch_chain_pass(false, a, s);
            return e;
        }
        if (has_model && kind == CONV) return launch_conv_kernel(a, cdesc, s);
        if (a.mode == MODE_CHAIN && chain_form() == 1) return launch_pipe_kernel(kernel, a, s);
        if (a.mode == MODE_CHAIN && chain_form() == 2) return launch_split_kernels(has_model ? kernel : nullptr, a, s);
        return launch_stream_kernel(has_model ? kernel : nullptr, a, lds_bytes(a.mode == MODE_CHAIN ? a.n_frames : 0), s);
    }
    void release()
    {
        if (d_ctl) (void)hipFree(d_ctl);
        if (d_st) (void)hipFree(d_st);
        if (d_nn) (void)hipFree(d_nn);
        if (d_wpack) (void)hipFree(d_wpack);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (q) (void)hipStreamDestroy(q);
        d_ctl = nullptr; d_st = nullptr; d_nn = nullptr; d_wpack = nullptr; d_in = nullptr; d_out = nullptr; q = nullptr;
    }
};

namespace {

void init_state(aidax_pool& p)
{
    // instantiate(), rt-neural-generic.cpp:283-321: preGain target 1 cleared, masterGain
    // target 0 cleared, biquad z = 0, no model, loading = true
    std::vector<StreamState> st(p.n_streams);
    std::memset(st.data(), 0, sizeof(StreamState) * p.n_streams);
    for (auto& s : st) {
        s.pre_mem = 1.f; s.pre_tgt = 1.f;
        s.master_mem = 0.f; s.master_tgt = 0.f;
    }
    HIP_TRY(hipMemcpyAsync(p.d_st, st.data(), sizeof(StreamState) * p.n_streams, hipMemcpyHostToDevice, p.q));
    HIP_TRY(hipStreamSynchronize(p.q));
}

int set_model_impl(aidax_pool& p, const aidax_model* m, int start_mode)
{
    HIP_TRY(hipSetDevice(p.device));
    if (!m) {
        p.has_model = false;
        p.kernel = nullptr;
        for (auto& l : p.loading) l = 1;
        p.refresh_all();
        return AIDAX_OK;
    }
    if (!model_supported(*m)) return fail(AIDAX_ERR_ARCH, "Unable to identify a known model architecture! (no kernel)");
    const KernelEntry* k = nullptr;
    std::vector<float> wp;
    uint32_t state_floats = 0;
    aidax_pool::Kind kind = aidax_pool::TABLE;
    StackDesc sd{};
    MfmaDesc md{};
    ConvDesc cd{};
    QuadDesc qd{};
    bool conv_mfma = false;
    if (is_conv_model(*m)) {
        kind = aidax_pool::CONV;
        wp = pack_conv(*m, &cd, &state_floats);
        conv_mfma = p.max_frames <= 256 && p.force_form != 4 && convm_lds_bytes(cd, p.max_frames) <= 160 * 1024;
        if (!conv_mfma && conv_lds_bytes(cd, p.max_frames) > 160 * 1024)
            return fail(AIDAX_ERR_ARG, "conv model: pool max_frames too large for the LDS activation planes");
    } else if (m->n_rnn == 1 && find_kernel(m->cell, m->hidden) && p.quad_for_table_model(m->cell, m->hidden) &&
               chain_lds_bytes(p.max_frames) <= kChainLdsLimit && quad_lds_bytes(m->hidden, p.max_frames) <= 160 * 1024) {
        kind = aidax_pool::QUAD;
        wp = pack_quad(*m, &qd.bias_off, &qd.dense_off);
        state_floats = static_cast<uint32_t>(m->cell == AIDAX_CELL_LSTM ? 2 * m->hidden : m->hidden);
    } else if (mfma_form_fits(*m) && chain_lds_bytes(p.max_frames) <= kChainLdsLimit &&
               (is_stack_model(*m) ? p.force_form != 4 : p.mfma_for_table_model(m->cell, m->hidden))) {
        kind = aidax_pool::MFMA;
        wp = pack_mfma(*m, &md, &state_floats);
    } else if (is_stack_model(*m)) {
        kind = aidax_pool::STACK;
        wp = pack_stack(*m, &sd, &state_floats);
        if (stack_lds_bytes(sd, p.max_frames) > 160 * 1024)
            return fail(AIDAX_ERR_ARG, "stacked model: pool max_frames too large for the LDS block buffers");
    } else {
        k = find_kernel(m->cell, m->hidden);
        wp = pack_weights(*m);
        if (static_cast<int>(wp.size()) != k->pack_regs * kWave + m->hidden + 1) return fail(AIDAX_ERR_STATE, "weight pack size mismatch");
        state_floats = static_cast<uint32_t>(k->state_floats);
    }

    // model swaps are rare (worker thread); drain everything that may still read the old buffers
    HIP_TRY(hipDeviceSynchronize());
    float* new_w = nullptr;
    float* new_nn = nullptr;
    HIP_TRY(hipMalloc(&new_w, wp.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(new_w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice, p.q));
    const uint32_t stride = (state_floats + 3u) & ~3u;
    HIP_TRY(hipMalloc(&new_nn, static_cast<size_t>(p.n_streams) * stride * sizeof(float)));
    HIP_TRY(hipStreamSynchronize(p.q));
    if (p.d_wpack) (void)hipFree(p.d_wpack);
    if (p.d_nn) (void)hipFree(p.d_nn);
    p.d_wpack = new_w;
    p.d_nn = new_nn;
    p.nn_stride = stride;
    p.kernel = k;
    p.kind = kind;
    p.sdesc = sd;
    p.mdesc = md;
    p.cdesc = cd;
    p.qdesc = qd;
    p.cell = m->cell;
    p.conv_mfma = conv_mfma;
    p.hidden = m->hidden;
    p.input_size = m->input_size;
    p.input_skip = m->input_skip;
    p.in_gain = m->input_gain;
    p.out_gain = m->output_gain;
    p.model_sr = m->samplerate;

    // fresh DynamicModel per stream: reset() + param smoothers around the inherited targets (:1035, :1053-1061)
    HIP_TRY(launch_reset_for_model(p.d_st, p.d_nn, p.n_streams, p.nn_stride, p.p_den(), p.q));
    p.has_model = true;
    if (start_mode == AIDAX_START_WARMUP) {           // 2048 zeros through applyModel (:1077-1078)
        const uint32_t chunk = p.launch_chunk(kWarmupFrames);
        for (uint32_t done = 0; done < kWarmupFrames; done += chunk) {
            LaunchArgs a = p.args(nullptr, nullptr, std::min(chunk, kWarmupFrames - done), MODE_WARMUP);
            HIP_TRY(p.launch(a, p.q));
        }
    }
    p.pipe_capacity = k ? pipe_resident_streams(k, p.max_frames, p.device) : 0;
    p.split_pays = k ? split_form_pays(k, p.max_frames) : false;
    p.has_model = true;
    for (auto& l : p.loading) l = 0;                  // work_response: loading = false (:889)
    p.refresh_all();
    return AIDAX_OK;
}

}  // namespace

extern "C" {

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
        if (const char* f = std::getenv("AIDAX_KERNEL"))
            p->force_form = std::strcmp(f, "wave") == 0 ? 1 : std::strcmp(f, "pipe") == 0 ? 2 : std::strcmp(f, "split") == 0 ? 3 : std::strcmp(f, "valu") == 0 ? 4 : std::strcmp(f, "mfma") == 0 ? 5 : std::strcmp(f, "quad") == 0 ? 6 : 0;
        try {
            HIP_TRY(hipSetDevice(device_id));
            HIP_TRY(hipStreamCreateWithFlags(&p->q, hipStreamNonBlocking));
            HIP_TRY(hipMalloc(&p->d_ctl, sizeof(StreamCtl) * n_streams));
            HIP_TRY(hipMalloc(&p->d_st, sizeof(StreamState) * n_streams));
            HIP_TRY(hipMalloc(&p->d_in, sizeof(float) * n_streams * static_cast<size_t>(max_frames)));
            HIP_TRY(hipMalloc(&p->d_out, sizeof(float) * n_streams * static_cast<size_t>(max_frames)));
            p->controls.resize(n_streams);
            for (auto& c : p->controls) aidax_controls_default(&c);
            p->loading.assign(n_streams, 1);
            p->h_ctl.resize(n_streams);
            p->refresh_all();
            init_state(*p);
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
    if (p->q) (void)hipStreamSynchronize(p->q);
    p->release();
    delete p;
}

AIDAX_API uint32_t aidax_pool_streams(const aidax_pool* p) { return p ? p->n_streams : 0; }

AIDAX_API int aidax_pool_set_model(aidax_pool* p, const aidax_model* m, int start_mode)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (start_mode != AIDAX_START_WARMUP && start_mode != AIDAX_START_RESET) return fail(AIDAX_ERR_ARG, "bad start_mode");
    return guarded([&]() { return set_model_impl(*p, m, start_mode); });
}

AIDAX_API int aidax_pool_reset_stream(aidax_pool* p, uint32_t stream, int start_mode)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (stream >= p->n_streams) return fail(AIDAX_ERR_ARG, "stream out of range");
    if (start_mode != AIDAX_START_WARMUP && start_mode != AIDAX_START_RESET) return fail(AIDAX_ERR_ARG, "bad start_mode");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        StreamState st{};                                   // instantiate(), :283-321
        st.pre_mem = 1.f; st.pre_tgt = 1.f;
        HIP_TRY(hipMemcpyAsync(p->d_st + stream, &st, sizeof(st), hipMemcpyHostToDevice, p->q));
        HIP_TRY(hipStreamSynchronize(p->q));                // `st` is a stack object
        if (p->has_model) {
            // a fresh DynamicModel for this stream only: the launch arguments view the pool as one stream
            HIP_TRY(launch_reset_for_model(p->d_st + stream, p->d_nn + static_cast<size_t>(stream) * p->nn_stride, 1,
                                           p->nn_stride, p->p_den(), p->q));
            if (start_mode == AIDAX_START_WARMUP) {
                const uint32_t chunk = p->launch_chunk(kWarmupFrames);
                for (uint32_t done = 0; done < kWarmupFrames; done += chunk) {
                    LaunchArgs a = p->args(nullptr, nullptr, std::min(chunk, kWarmupFrames - done), MODE_WARMUP);
                    a.ctl += stream; a.st += stream; a.nn += static_cast<size_t>(stream) * p->nn_stride;
                    a.n_streams = 1;
                    p->flush_ctl(p->q);
                    HIP_TRY(p->launch(a, p->q));
                }
            }
        }
        p->loading[stream] = p->has_model ? 0 : 1;
        p->refresh_ctl(stream);
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_set_loading(aidax_pool* p, int32_t stream, int loading)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (stream != AIDAX_ALL_STREAMS && (stream < 0 || static_cast<uint32_t>(stream) >= p->n_streams))
        return fail(AIDAX_ERR_ARG, "stream out of range");
    if (stream == AIDAX_ALL_STREAMS) {
        for (auto& l : p->loading) l = loading ? 1 : 0;
        p->refresh_all();
    } else {
        p->loading[stream] = loading ? 1 : 0;
        p->refresh_ctl(static_cast<uint32_t>(stream));
    }
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
        // paramFirstRun is only re-armed when a model exists (rt-neural-generic.cpp:344-351)
        const uint32_t bits = PEND_ACTIVATE | (p->has_model ? PEND_PARAM_FIRST : 0u);
        HIP_TRY(launch_set_pending(p->d_st, p->n_streams, stream, bits, p->q));
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_process_device(aidax_pool* p, const float* d_in, float* d_out, uint32_t n_frames, void* hip_stream)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (n_frames > p->max_frames) return fail(AIDAX_ERR_ARG, "n_frames exceeds the pool's max_frames");
    if (n_frames != 0 && (!d_in || !d_out)) return fail(AIDAX_ERR_ARG, "null buffer");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : p->q;
        p->flush_ctl(s);
        LaunchArgs a = p->args(d_in, d_out, n_frames, MODE_CHAIN);
        HIP_TRY(p->launch(a, s));
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_process(aidax_pool* p, const float* in, float* out, uint32_t n_frames)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    if (n_frames > p->max_frames) return fail(AIDAX_ERR_ARG, "n_frames exceeds the pool's max_frames");
    if (n_frames != 0 && (!in || !out)) return fail(AIDAX_ERR_ARG, "null buffer");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        const size_t bytes = sizeof(float) * p->n_streams * static_cast<size_t>(n_frames);
        if (bytes) HIP_TRY(hipMemcpyAsync(p->d_in, in, bytes, hipMemcpyHostToDevice, p->q));
        const int rc = aidax_pool_process_device(p, p->d_in, p->d_out, n_frames, p->q);
        if (rc != AIDAX_OK) return rc;
        if (bytes) HIP_TRY(hipMemcpyAsync(out, p->d_out, bytes, hipMemcpyDeviceToHost, p->q));
        HIP_TRY(hipStreamSynchronize(p->q));
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_sync(aidax_pool* p)
{
    if (!p) return fail(AIDAX_ERR_ARG, "null pool");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(hipStreamSynchronize(p->q));
        return AIDAX_OK;
    });
}

AIDAX_API int aidax_pool_read_state(aidax_pool* p, uint32_t stream, int layer, float* h, float* c, uint32_t cap)
{
    if (!p || !h) return fail(AIDAX_ERR_ARG, "null argument");
    if (!p->has_model || stream >= p->n_streams || p->kind == aidax_pool::CONV) return fail(AIDAX_ERR_STATE, "no such state");
    const int n_layers = p->kind == aidax_pool::TABLE ? 1 : p->kind == aidax_pool::MFMA ? p->mdesc.n_layers : p->sdesc.n_layers;
    if (layer < 0 || layer >= n_layers) return fail(AIDAX_ERR_STATE, "no such layer");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(hipStreamSynchronize(p->q));
        uint32_t H = static_cast<uint32_t>(p->hidden), off = 0;
        bool lstm = false;
        if (p->kind == aidax_pool::STACK) {
            H = static_cast<uint32_t>(p->sdesc.L[layer].hidden); off = p->sdesc.L[layer].state_off; lstm = p->sdesc.L[layer].cell == 0;
        } else if (p->kind == aidax_pool::MFMA) {
            H = static_cast<uint32_t>(p->mdesc.hidden_true); off = p->mdesc.L[layer].state_off; lstm = p->mdesc.L[layer].cell == 0;
        } else {
            lstm = p->cell == AIDAX_CELL_LSTM;
        }
        const uint32_t n = H < cap ? H : cap;
        const float* base = p->d_nn + static_cast<size_t>(stream) * p->nn_stride + off;
        HIP_TRY(hipMemcpy(h, base, n * sizeof(float), hipMemcpyDeviceToHost));
        if (c && lstm) HIP_TRY(hipMemcpy(c, base + H, n * sizeof(float), hipMemcpyDeviceToHost));
        return static_cast<int>(H);
    });
}

AIDAX_API const char* aidax_pool_kernel_name(const aidax_pool* p)
{
    if (!(p && p->has_model)) return "k_nomodel";
    if (p->kind == aidax_pool::STACK) return "k_stack";
    if (p->kind == aidax_pool::MFMA) return "k_chain+k_mfma";
    if (p->kind == aidax_pool::QUAD) return "k_chain+k_quad";
    if (p->kind == aidax_pool::CONV) return p->conv_mfma ? "k_chain+k_conv_mfma" : "k_conv";
    const int form = p->chain_form();
    return form == 1 ? p->kernel->name_pipe : form == 2 ? p->kernel->name_split : p->kernel->name;
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
            const uint32_t chunk = p->launch_chunk(n ? n : 1);
            for (uint32_t done = 0; done < n && le == hipSuccess; done += chunk) {
                LaunchArgs a = p->args(d_x + static_cast<size_t>(done) * m->input_size, d_y + done,
                                       std::min(chunk, n - done), MODE_NN_ONLY);
                if (unit_gains) { a.in_gain = 1.f; a.out_gain = 1.f; }
                le = p->launch(a, p->q);
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
