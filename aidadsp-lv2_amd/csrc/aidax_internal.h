// aidax_internal.h — host-side types behind the C ABI of include/aidax.h.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/aidax.h"
#include "aidax_layout.h"

namespace aidax {

struct Layer {
    enum Type { LSTM, GRU, DENSE, CONV1D } type;
    int in_size = 0, out_size = 0;
    int ksize = 0, dilation = 0;
    int activation = 0;                       // 0 none, 1 tanh, 2 relu, 3 sigmoid
    std::vector<float> w0, w1, w2;            // Keras layouts, see include/aidax.h / aidax_model.cpp
};

}  // namespace aidax

// The opaque model of the C ABI: everything loadModelFromPath reads from the
// json (rt-neural-generic.cpp:977-1013) plus the layer weights parseJson would
// have pulled into RTNeural (:1034).
struct aidax_model {
    std::string path;
    int cell = 0, hidden = 0, input_size = 0, n_rnn = 0;
    int input_skip = 0;
    float input_gain = 1.f, output_gain = 1.f, samplerate = 48000.f;
    bool in_reference_set = false;
    std::vector<aidax::Layer> layers;
    std::vector<float> golden_in, golden_out;
    uint64_t n_weights = 0;
};

namespace aidax {

void set_error(const std::string& msg);
int  fail(int code, const std::string& msg);

// host DSP helpers (aidax_dsp_host.cpp)
void  design_biquad(int type, double fc, double q, double gain_db, double out[5]);
float db_to_coeff(float db);
float lpf_fc(float percent);
float exp_smoother_coef(float samplerate, float t60);
// controls + instance flags -> the per-stream record the kernels read
void  build_stream_ctl(const aidax_controls& c, double host_samplerate, bool has_model, bool loading,
                       float gain_coef, float p_den, StreamCtl* out);

// aidax_pool_process_device restricted to the first n_active streams (aidax_pool.cpp; used by the hub)
int pool_process_prefix(aidax_pool* p, const float* d_in, float* d_out, uint32_t n_frames, void* hip_stream, uint32_t n_active);

// park / unpark a stream (behaves disabled while parked; its controls stay what they are) — aidax_pool.cpp, used by the hub
int pool_park_stream(aidax_pool* p, uint32_t stream, bool parked);

// a stream that continues another plugin instance's life (hub mode; aidax_pool.cpp)
int  pool_reset_stream_inherit(aidax_pool* p, uint32_t stream, int start_mode, const float* p_targets);
int  pool_adopt_stream_dsp(aidax_pool* dst, uint32_t ds, aidax_pool* src, uint32_t ss);
int  pool_read_stream_state(aidax_pool* p, uint32_t stream, StreamState* out);
bool pool_has_model(const aidax_pool* p);
int  pool_peek_stream_state(aidax_pool* p, uint32_t stream, StreamState* pinned_out, void* done_event);

// k_mfma_lp fault report of a pool (aidax_pool.cpp): true once per give-up; the pool then serves its model with k_mfma
bool pool_take_lp_fault(aidax_pool* p);
bool pool_chained_kernel_in_use(aidax_pool* p);      // the next pass of the playing model goes out on k_mfma_lp / k_mfma_ls (whose hand-over can give up)
bool pool_lp_in_use(const aidax_pool* p);

// weight packing (aidax_pack.cpp)
std::vector<float> pack_weights(const aidax_model& m);
// extension architectures: flat weight buffer + descriptor + per-stream state size (floats)
bool is_stack_model(const aidax_model& m);     // >= 2 recurrent layers, or one layer wider than the register-resident kernels
bool is_conv_model(const aidax_model& m);
std::vector<float> pack_stack(const aidax_model& m, StackDesc* d, uint32_t* state_floats);
bool mfma_form_fits(const aidax_model& m);    // recurrent layers of one width, a multiple of 16 and <= 128
std::vector<float> pack_mfma(const aidax_model& m, MfmaDesc* d, uint32_t* state_floats);
void split_bf16x3(float x, uint16_t (&terms)[3]);     // x = t0 + t1 + t2 exactly, each a bf16 (k_gru_gs's weights; the kernel splits h the same way)
std::vector<float> pack_quad(const aidax_model& m, uint32_t* bias_off, uint32_t* dense_off);   // table models only
std::vector<float> pack_q4(const aidax_model& m);       // LSTM-32, one input: the record k_lstm_q4 reads
std::vector<float> pack_conv(const aidax_model& m, ConvDesc* d, uint32_t* state_floats);
int  conv_ms_tap(int ksize, int pos);
bool conv_ms_shape_ok(const ConvDesc& d);
int conv_st_shape(const ConvDesc& d);          // 1 + the index of the stack's k_conv_st geometry (aidax_layout.h), 0 = none
uint32_t conv_ms_period(const ConvDesc& d);

}  // namespace aidax
