// aidax_kernels.h — host-callable launchers defined in aidax_kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include "aidax_layout.h"

namespace aidax {

struct KernelEntry {
    int cell, hidden;
    void (*fn)(LaunchArgs);           // one wavefront per stream (throughput form)
    void (*fn_pipe)(LaunchArgs);      // three wavefronts per stream (latency form)
    void (*fn_nn)(LaunchArgs);        // recurrent cell only, between the two packed chain launches (split form)
    int pack_regs, state_floats;
    const char* name;
    const char* name_pipe;
    const char* name_split;
    void (*fn_pipe4)(LaunchArgs);     // four streams per workgroup, one helper wave (k_*_pipe4; nullptr: the cell has none)
    const char* name_pipe4;
    void (*fn_pipe4c)(LaunchArgs);    // the same for a conditioned model (PARAM1 / PARAM2 as inputs): the helper wave also runs the PARAM smoothers
};

const KernelEntry* find_kernel(int cell, int hidden);
hipError_t launch_stream_kernel(const KernelEntry* e, const LaunchArgs& a, size_t lds_bytes, hipStream_t stream);
hipError_t launch_split_kernels(const KernelEntry* e, const LaunchArgs& a, hipStream_t stream);
bool split_form_pays(const KernelEntry* e, uint32_t n_frames);
size_t pipe_lds_bytes(int hidden, uint32_t n_frames);
hipError_t launch_pipe_kernel(const KernelEntry* e, const LaunchArgs& a, hipStream_t stream, hipEvent_t done = nullptr);
int pipe_resident_streams(const KernelEntry* e, uint32_t n_frames, int device);
// k_*_pipe4: a pass of whole 16-frame tiles, whole workgroups of four streams, (the caller checks)
size_t pipe4_lds_bytes(int hidden, uint32_t n_frames, int input_size);
hipError_t launch_pipe4_kernel(const KernelEntry* e, const LaunchArgs& a, hipStream_t stream, hipEvent_t done = nullptr);
size_t stack_lds_bytes(const StackDesc& d, uint32_t n_frames);
size_t conv_lds_bytes(const ConvDesc& d, uint32_t n_frames);
size_t mfma_lds_bytes(const MfmaDesc& d, uint32_t n_frames);
size_t chain_lds_bytes(uint32_t n_frames);
constexpr size_t kChainLdsLimit = 72 * 1024;     // packed chains: eight rows of <= 2048 frames + hand-over slots
// one packed chain launch (8 streams per wave): pre = LPF/pre-gain/EQ(pre) in -> out, else DC/EQ(post)/master in place
hipError_t launch_chain_pass(bool pre, const LaunchArgs& a, hipStream_t stream);
hipError_t launch_mfma_kernel(const LaunchArgs& a, const MfmaDesc& d, hipStream_t stream);
// k_mfma_lp (aidax_mfmalp.hip): stacked models, one workgroup per (16 streams, layer), layers chained through a global ring
bool mfma_lp_serves(const MfmaDesc& d);
size_t mfma_lp_lds_bytes(const MfmaDesc& d, uint32_t n_frames, bool fused = false);
bool mfma_lp_fused_serves(const MfmaDesc& d, uint32_t max_frames);       // the DSP chain inside the same launch: one-layer models (helper waves), stacked models with blocks <= 256 frames
size_t mfma_lp_ring_bytes(const MfmaDesc& d, uint32_t n_streams);
size_t mfma_lp_counter_bytes(const MfmaDesc& d, uint32_t n_streams);
// `fault`: device view of a word in pinned host memory that a workgroup bumps when a hand-over wait timed out
// k_gru_gm (aidax_mfmalp.hip): one-layer GRU on gate-major tiles, the whole run() in one launch (MODE_CHAIN, n_frames > 0)
bool gru_gm_serves(const MfmaDesc& d);
size_t gru_gm_lds_bytes(const MfmaDesc& d, uint32_t n_frames);
hipError_t launch_gru_gm_kernel(const LaunchArgs& a, const MfmaDesc& d, hipStream_t stream);
// k_gru_gs (aidax_mfmalp.hip): k_gru_gm with the recurrent product as bf16 MFMAs of operands split exactly into three bf16
// terms; n_products: 6 (to fp32 rounding) or 9 (every term product)
bool gru_gs_serves(const MfmaDesc& d);
size_t gru_gs_lds_bytes(const MfmaDesc& d, uint32_t n_frames);
hipError_t launch_gru_gs_kernel(const LaunchArgs& a, const MfmaDesc& d, int n_products, hipStream_t stream);
// k_lstm_gs (aidax_mfmalp.hip): the same structure for one-layer LSTMs of 40 (run as 48) / 64 units (pack_mfma's LSTM record at gs_off)
bool lstm_gs_serves(const MfmaDesc& d);
size_t lstm_gs_lds_bytes(const MfmaDesc& d, uint32_t n_frames);
hipError_t launch_lstm_gs_kernel(const LaunchArgs& a, const MfmaDesc& d, int n_products, hipStream_t stream);
hipError_t launch_mfma_lp_kernel(const LaunchArgs& a, const MfmaDesc& d, float* ring, uint32_t* counters, uint32_t* fault, hipStream_t stream, bool fused = false);
// k_mfma_ls (aidax_mfmalp.hip): k_mfma_lp's stacked models with the contractions as bf16 MFMAs of operands split exactly into
// three bf16 terms (n_products: 6 or 9); same ring protocol, its own ring geometry (a frame is the h fragments)
bool mfma_ls_serves(const MfmaDesc& d);
size_t mfma_ls_lds_bytes(const MfmaDesc& d, uint32_t n_frames, bool fused = false);
bool mfma_ls_fused_serves(const MfmaDesc& d, uint32_t max_frames);
size_t mfma_ls_ring_bytes(const MfmaDesc& d, uint32_t n_streams);
hipError_t launch_mfma_ls_kernel(const LaunchArgs& a, const MfmaDesc& d, float* ring, uint32_t* counters, uint32_t* fault, int n_products,
                                 hipStream_t stream, bool fused = false);
// k_lstm_q4 (aidax_q4.hip): LSTM-32 snapshot models, four streams per workgroup, the whole run() in one launch
bool q4_serves(int cell, int hidden, int input_size);
size_t q4_lds_bytes(int hidden, uint32_t n_frames);
hipError_t launch_q4_kernel(int hidden, const LaunchArgs& a, hipStream_t stream);     // a.wpack = the pack_q4 record
size_t quad_lds_bytes(int hidden, uint32_t n_frames);
hipError_t launch_quad_kernel(int cell, int hidden, const LaunchArgs& a, const QuadDesc& qd, hipStream_t stream);
hipError_t launch_stack_kernel(const LaunchArgs& a, const StackDesc& d, hipStream_t stream);
size_t convm_lds_bytes(const ConvDesc& d, uint32_t n_frames);
int convm_resident_streams(const ConvDesc& d, uint32_t n_frames, int device);                 // workgroups of the fused form resident at once
hipError_t launch_conv_mfma_kernel(const LaunchArgs& a, const ConvDesc& d, bool fused, hipStream_t stream);   // n_frames <= 256; fused: whole run()
hipError_t launch_conv_kernel(const LaunchArgs& a, const ConvDesc& d, hipStream_t stream);
// k_conv_ms (aidax_convs.hip): the conv stacks conv_ms_shape_ok admits, as bf16 term products; its own history layout (ConvDesc::ms_*)
size_t convs_lds_bytes();
int convs_resident_streams(int device, int st_geo);      // st_geo: the stack's k_conv_st geometry (ConvDesc::st_ok - 1), -1 = none
bool conv_st_block_ok(uint32_t n_frames);                // a block length k_conv_st is compiled for (64 / 128 / 256 frames)
hipError_t launch_conv_ms_kernel(const LaunchArgs& a, const ConvDesc& d, bool fused, hipStream_t stream);   // n_frames <= 256; fused: whole run()
hipError_t launch_set_pending(StreamState* st, uint32_t n_streams, int32_t stream, uint32_t bits, hipStream_t q);
hipError_t launch_init_streams(StreamState* st, uint32_t n, hipStream_t q);
hipError_t launch_stage_params(const StreamState* live, StreamState* staged, uint32_t n, hipStream_t q);
hipError_t launch_install_params(StreamState* live, const StreamState* staged, uint32_t n, hipStream_t q);
hipError_t launch_set_param_targets(StreamState* st, float t0, float t1, hipStream_t q);
hipError_t launch_adopt_dsp(StreamState* dst, const StreamState* src, hipStream_t q);
hipError_t launch_reset_for_model(StreamState* st, float* nn, uint32_t n_streams, uint32_t nn_stride, float p_den, hipStream_t q);

hipError_t launch_keep_warm_kernel(int workgroups, hipStream_t stream);      // an empty grid (AIDAX_KEEP_WARM_US, aidax_pool.cpp)

}  // namespace aidax
