/*
 * aidax.h — C ABI of the MI355X-native rt-neural-generic hot path.
 *
 * This is the drop-in boundary: the reference's LV2 shell keeps its own
 * plumbing (ports, atoms, worker, state) and calls these entry points where it
 * used to call its four private DSP statics (rt-neural-generic.h:321-324) and
 * its model loader. Every entry point cites the reference code it replaces
 * (paths relative to the reference repo root). Plain pointers and sizes only;
 * integer status codes; no exceptions cross this boundary; `process` never
 * allocates. There is NO CPU fallback: without a usable HIP device every
 * device-touching call fails with AIDAX_ERR_DEVICE.
 *
 * One `aidax_pool` = N independent mono streams (N plugin instances' DSP
 * state) sharing one model, resident on one GPU. The reference processes one
 * stream per plugin instance; a host that wants GPU batching aggregates its
 * instances into one pool. The LV2 shell built from
 * aidadsp-lv2_amd/lv2/ uses a pool of N=1 per instance.
 */
#ifndef AIDAX_H
#define AIDAX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define AIDAX_API __attribute__((visibility("default")))
#else
#define AIDAX_API
#endif

/* ------------------------------------------------------------------ status */
enum {
    AIDAX_OK = 0,
    AIDAX_ERR_ARG = -1,        /* null / out-of-range argument                                  */
    AIDAX_ERR_IO = -2,         /* file unreadable                                               */
    AIDAX_ERR_JSON = -3,       /* json malformed or a required key missing ("Unable to load
                                  json file", rt-neural-generic.cpp:1017-1020)                  */
    AIDAX_ERR_ARCH = -4,       /* "Unable to identify a known model architecture!" (:1025-1026),
                                  input_size > 3 (:978-980) or in_skip > 1 (:984-985)           */
    AIDAX_ERR_DEVICE = -5,     /* HIP error / no device / kernels not loadable                  */
    AIDAX_ERR_STATE = -6       /* call not valid in the current state                           */
};

AIDAX_API const char* aidax_last_error(void);      /* thread-local message of the last failure */
AIDAX_API const char* aidax_version(void);

/* ------------------------------------------------------------------- model */
typedef struct aidax_model aidax_model;            /* host-side DynamicModel minus its state
                                                      (rt-neural-generic.h:115-129)             */

enum { AIDAX_CELL_LSTM = 0, AIDAX_CELL_GRU = 1, AIDAX_CELL_CONV = 2 };

typedef struct {
    int32_t cell;              /* AIDAX_CELL_*  (layers[0].type, model_variant.hpp:62-71)       */
    int32_t hidden;            /* layers[0].shape[-1]                                           */
    int32_t input_size;        /* in_shape[-1] in 1..3 (rt-neural-generic.cpp:977-980); this is
                                  also the value of the ModelInSize output port (:518)          */
    int32_t n_rnn_layers;      /* 1 for every architecture the reference can load; >1 or cell ==
                                  CONV are extensions                           */
    int32_t input_skip;        /* in_skip (:982-989)                                            */
    float   input_gain;        /* DB_CO(in_gain) or 1 (:991-996)                                */
    float   output_gain;       /* DB_CO(out_gain) or 1 (:998-1003)                              */
    float   samplerate;        /* metadata.samplerate / samplerate if NUMBER, else 48000 (:1005-1013) */
    int32_t n_golden;          /* length of input_batch/output_batch if both present, else 0    */
    int32_t in_reference_set;  /* 1 if one of the 54 variants of model_variant.hpp:6-59         */
    uint64_t n_weights;
} aidax_model_info_t;

/* loadModelFromPath, parse + architecture match half (rt-neural-generic.cpp:963-1044).
 * Host only; does not touch the GPU. Besides the reference's 54 variants this build loads its extension
 * architectures (n_rnn_layers > 1, wider single layers, conv1d stacks; in_reference_set == 0); with
 * AIDAX_STRICT_REFERENCE_SET=1 in the environment those are rejected with the reference's own message
 * (:1025-1026). The LV2 shell applies that rule by default (AIDAX_STRICT_REFERENCE_SET=0 lifts it). */
AIDAX_API int  aidax_model_load(const char* json_path, aidax_model** out);
AIDAX_API int  aidax_model_load_memory(const char* json_text, size_t len, const char* label, aidax_model** out);
AIDAX_API int  aidax_model_info(const aidax_model* m, aidax_model_info_t* info);
AIDAX_API const char* aidax_model_path(const aidax_model* m);            /* DynamicModel::path */
/* copies min(cap, n_golden) floats of input_batch / output_batch (either may be NULL) */
AIDAX_API int  aidax_model_golden(const aidax_model* m, float* in, float* out, uint32_t cap);
AIDAX_API void aidax_model_free(aidax_model* m);                          /* freeModel :1093-1101 */

/* ---------------------------------------------------------------- controls */
/* The 20 input control ports of the generic build, by value, in TTL order
 * (ports_t rt-neural-generic.h:84-112; ranges/defaults rt-neural-generic.ttl:94-313). */
typedef struct {
    float in_lpf_pc;           /*  4 ANTIALIASING %   */
    float pregain_db;          /*  5 PREGAIN dB       */
    float net_bypass;          /*  6 NETBYPASS        */
    float param1;              /*  7 PARAM1           */
    float param2;              /*  8 PARAM2           */
    float eq_bypass;           /*  9 EQBYPASS         */
    float eq_position;         /* 10 EQPOS 0=post 1=pre */
    float bass_boost_db;       /* 11 BASS             */
    float bass_freq;           /* 12 BFREQ            */
    float mid_boost_db;        /* 13 MID              */
    float mid_freq;            /* 14 MFREQ            */
    float mid_q;               /* 15 MIDQ             */
    float mid_type;            /* 16 MTYPE 0=peak 1=bandpass */
    float treble_boost_db;     /* 17 TREBLE           */
    float treble_freq;         /* 18 TFREQ            */
    float depth_boost_db;      /* 19 DEPTH            */
    float presence_boost_db;   /* 20 PRESENCE         */
    float dc_blocker;          /* 21 DCBLOCKER        */
    float master_db;           /* 22 MASTER           */
    float enabled;             /* 24 enabled          */
} aidax_controls;

AIDAX_API void aidax_controls_default(aidax_controls* c);                /* lv2:default values */

/* Biquad::setBiquad -> calcBiquad (common/Biquad.cpp:60-165): coefficients
 * a0,a1,a2,b1,b2 for (type 0..6, Fc/fs, Q, gain dB). Host only. */
AIDAX_API int aidax_biquad_design(int type, double fc, double q, double gain_db, double coeffs[5]);
/* DB_CO (rt-neural-generic.h:160) and the ANTIALIASING % -> Fc map (.h:167,178-179; cpp:515) */
AIDAX_API float aidax_db_to_coeff(float db);
AIDAX_API float aidax_lpf_fc(float percent);

/* ------------------------------------------------------- device placement
 * The reference's unit of independence is the plugin instance (instantiate(), rt-neural-generic.cpp:244-333; instances
 * share nothing, rt-neural-generic.h:198-239, 311-319), so on a node with several GPUs instances are the thing to spread.
 * aidax_device_count: HIP devices this process sees (0 and AIDAX_ERR_DEVICE without a usable runtime).
 * aidax_pick_device: the placement rule, a pure function (no HIP call): `spec` names the candidates — NULL / "" / "0" the
 * first device (what AIDAX_DEVICE unset means), "auto" every device, else a list of indices and ranges ("0-3,6"); of the
 * candidates below device_count the one with the smallest load[] wins, ties go to the lowest index. load may be NULL (all
 * idle). AIDAX_ERR_ARG when spec is malformed or names no device below device_count. The LV2 shell keeps one load count
 * per device and process (instances in one-stream mode, hubs' seats in hub mode) and asks this function at instantiate()
 * resp. when it opens a hub. */
AIDAX_API int aidax_device_count(int* count);
AIDAX_API int aidax_pick_device(const char* spec, int device_count, const uint32_t* load, int* device_out);
/* aidax_pick_hub: which of the n hubs that serve one model file an instance joins (hub mode of the LV2 shell), a pure function too.
 * hub_device[i] / hub_free_seats[i] describe hub i; current_device is the device of the hub the instance plays in NOW (a model-file
 * swap: work() -> work_response(), rt-neural-generic.cpp:807-893), or -1 for an instance that joins its first hub. An instance
 * that already plays STAYS ON ITS DEVICE — its seven biquad memories and both gain smoothers are carried into the new seat by a
 * device-side copy (aidax_hub_adopt; the reference keeps them across a swap, :868-875), which two pools on different devices
 * cannot do. *index_out: the first hub on that device with a free seat, or -1 = open a new hub, on *device_out. For a first join:
 * the first hub with a free seat whatever its device; else a new hub on the device aidax_pick_device(spec, ...) names.
 * AIDAX_ERR_ARG as for aidax_pick_device, or when current_device is not below device_count. */
AIDAX_API int aidax_pick_hub(const int* hub_device, const uint32_t* hub_free_seats, int n_hubs, int current_device,
                             const char* spec, int device_count, const uint32_t* load, int* index_out, int* device_out);

/* ------------------------------------------------------- launch-form rule (diagnostic)
 * Which kernel family a pool of n_streams instances of a ONE-layer model of the reference's table (cell: AIDAX_CELL_LSTM /
 * AIDAX_CELL_GRU, hidden units) takes on a device with compute_units CUs, beyond the per-stream forms: 0 none (the wavefront /
 * pipeline / split forms, chosen by residency at pool creation), 1 four streams per workgroup on the 4x4 matrix instruction
 * (k_quad), 2 sixteen streams per workgroup on the 16x16 ones (k_gru_gs, k_lstm_gs, k_mfma_ls1, k_mfma_lp, k_mfma). A pure
 * function (no HIP call, environment switches apply): the decision table of csrc/aidax_pool.cpp, measured at 256-frame blocks on
 * 256 CUs; exported so that hosts and tests can see what a pool size will run on. */
AIDAX_API int aidax_many_streams_form(int cell, int hidden, uint32_t n_streams, int compute_units);
/* ... for a pool made for blocks of max_frames frames (the reference's run() is called with the HOST's period, rt-neural-generic.cpp:484:
 * 64 or 128 frames on MOD devices and low-latency set-ups). The table was re-measured at 64 / 128 / 256 frames (profiles/r05_blocklen_forms*.txt):
 * one crossover moves with the block length (LSTM-32 at 5632 .. 6144 streams); aidax_many_streams_form is this function at 256 frames. */
AIDAX_API int aidax_many_streams_form_at(int cell, int hidden, uint32_t n_streams, int compute_units, uint32_t max_frames);

/* ... and which kernel family a conv1d-stack model (an extension of the path: BASELINE config 4) takes, by its shape alone — the packer's
 * rules (csrc/aidax_pack.cpp: conv_ms_shape_ok, conv_st_shape_ok), pure, no HIP call: 0 not a conv model, 1 k_conv (VALU, any stack the
 * loader admits), 2 k_conv_mfma (fp32 matrix instructions: up to sixteen equal channels, a plane that fits LDS), 3 k_conv_ms (bf16 term
 * products: exactly sixteen channels, two to four taps, histories its plane + two register fragments hold), 4 k_conv_ms with FULL 256-frame
 * blocks of the fused form on k_conv_st, the streaming kernel (eight layers of three taps, dilation 2^l). Environment switches of the test
 * build are not applied: this is the shape rule. */
AIDAX_API int aidax_model_conv_form(const aidax_model* m);

/* -------------------------------------------------------------------- pool */
typedef struct aidax_pool aidax_pool;

enum { AIDAX_ALL_STREAMS = -1 };
enum {
    AIDAX_START_WARMUP = 0,    /* release build: 2048-zero pre-buffer through applyModel after
                                  reset() (rt-neural-generic.cpp:1075-1079)                     */
    AIDAX_START_RESET = 1      /* reset() state only (what the DEBUG self-test path leaves)     */
};

/* instantiate(), DSP half (rt-neural-generic.cpp:283-321) for n_streams
 * instances on GPU `device_id`: gain smoothers (T60 0.1 s at host rate, pre
 * target 1, master target 0), the seven biquads, loading = true, no model.
 * max_frames bounds n_frames of every later process call (<= 8192 for the reference's model table; the
 * extension architectures stage whole blocks in LDS and take max_frames <= 2048 (stacked / wide
 * recurrent) resp. <= ~1000 (conv stacks): aidax_pool_set_model reports AIDAX_ERR_ARG beyond that). */
AIDAX_API int  aidax_pool_create(uint32_t n_streams, uint32_t max_frames, double host_samplerate,
                                 int device_id, aidax_pool** out);
AIDAX_API void aidax_pool_destroy(aidax_pool* p);                         /* cleanup() :664-677 */
AIDAX_API uint32_t aidax_pool_streams(const aidax_pool* p);

/* Model swap, split the way the reference splits it between its two threads (:807-893).
 *
 * aidax_pool_prepare_model — work() / loadModelFromPath's device half (:822-836, :1034-1079), WORKER thread:
 *   packs the weights for the kernel form this pool will use, allocates and uploads them, gives every stream
 *   a fresh DynamicModel (reset(), PARAM smoothers rebuilt around the targets the playing model holds at that
 *   moment, :822-825 and :1053-1061) and runs the warm-up per start_mode — all into buffers of its own, on a
 *   stream of its own, while the audio thread keeps playing the old model. Blocks until the model is complete.
 *   model == NULL prepares an unload (no model, loading = true). The caller still owns `m`.
 * aidax_pool_commit_model — work_response() (:859-893), AUDIO thread: swaps the prepared model in and clears
 *   `loading`. Host assignments plus one small kernel on the pool's stream: no allocation, no free, no wait of
 *   any kind. After the call `staged` holds what the swap retired (the previous model's buffers);
 * aidax_staged_free — the reference's kWorkerFree leg (:838-840, :868-875), WORKER thread: frees a staged
 *   object (the retired buffers after a commit; the unused model if it was never committed).
 * aidax_pool_set_model is prepare + commit + free in one blocking call, for hosts without a worker. */
typedef struct aidax_staged aidax_staged;
AIDAX_API int  aidax_pool_prepare_model(aidax_pool* p, const aidax_model* m, int start_mode, aidax_staged** out);
AIDAX_API int  aidax_pool_commit_model(aidax_pool* p, aidax_staged* staged);
AIDAX_API void aidax_staged_free(aidax_staged* staged);
AIDAX_API int  aidax_pool_set_model(aidax_pool* p, const aidax_model* m, int start_mode);

/* Threads. A pool is driven by ONE audio-side caller at a time (set_controls, set_loading, activate,
 * reset_stream, commit_model, process*, sync) plus, concurrently, ONE worker-side caller (prepare_model,
 * staged_free). None of the audio-side calls allocates or frees device or pinned memory, and only
 * aidax_pool_process / aidax_pool_sync wait for the GPU (for the stream that carries the pass, never for the
 * device). */

/* One stream becomes a fresh plugin instance with the pool's model: instantiate() state for its DSP
 * members (:283-321; gain smoothers pre = 1 / master = 0 cleared, biquad states 0) and a fresh DynamicModel
 * (:1035-1079: reset, PARAM smoothers rebuilt, warm-up per start_mode), loading cleared if the pool has a
 * model. Used when a host attaches a new instance to a running pool (aidax_hub below). */
AIDAX_API int  aidax_pool_reset_stream(aidax_pool* p, uint32_t stream, int start_mode);

/* The plugin-owned DSP members of one stream — what RtNeuralGeneric keeps in itself rather than in its DynamicModel
 * (rt-neural-generic.h:311-317: seven Biquads, preGain, masterGain) and therefore keeps across a model swap (:868-875) —
 * plus the PARAM targets of the playing model, which work() hands to the next one (:822-825). */
typedef struct {
    double z[7][2];            /* Biquad z1, z2 (common/Biquad.h:50): in_lpf, dc_blocker, depth, bass, mid, treble, presence */
    float  pre_mem, master_mem, pre_target, master_target;     /* ExponentialValueSmoother mem / target as last set */
    float  param_target[2];    /* LinearValueSmoother targets of the playing model (0 / 0 without one) */
} aidax_stream_dsp;

/* Read / write those members of one stream. Both wait for the pool's passes (worker / main thread, tests); the
 * audio-thread way of moving a plugin instance between pools is aidax_hub_adopt below. import leaves the PARAM
 * smoothers alone (they belong to the pool's model) and, like a model swap, arms no activate(). */
AIDAX_API int  aidax_pool_export_stream_dsp(aidax_pool* p, uint32_t stream, aidax_stream_dsp* out);
AIDAX_API int  aidax_pool_import_stream_dsp(aidax_pool* p, uint32_t stream, const aidax_stream_dsp* in);

/* The `loading` flag (:318, :576, :889): while set, the master gain target is 0. */
AIDAX_API int  aidax_pool_set_loading(aidax_pool* p, int32_t stream, int loading);

/* Latch the control-port values run() reads at :489-502 for one stream or
 * AIDAX_ALL_STREAMS. Biquad coefficients are recomputed on the host exactly
 * where the reference's change detection would (:68-127, :514-517). */
AIDAX_API int  aidax_pool_set_controls(aidax_pool* p, int32_t stream, const aidax_controls* c);

/* activate() (:337-351): clear both gain smoothers to their current targets and
 * re-arm paramFirstRun. Recurrent state is NOT reset (the reference's reset is #if 0). */
AIDAX_API int  aidax_pool_activate(aidax_pool* p, int32_t stream);

/* run(), audio half (:607-659), for all streams: `in`/`out` are host buffers
 * laid out [n_streams][n_frames] (in-place allowed). Blocking: waits for the pool's stream. n_frames == 0 is
 * the legal "pre-run" (:606-609) and only latches targets. The block travels through pinned staging that the
 * pool allocated at creation (blocks of <= 64 KiB — the one-instance plugin — are read and written by the
 * kernels in place in pinned host memory, no copy engine involved; AIDAX_ZEROCOPY=0 turns that off — and such
 * a pool learns of the pass's end from a word the stream writes into pinned host memory, which the caller polls:
 * no interrupt and no wake-up on the way back; AIDAX_SPIN_WAIT=0 waits with hipStreamSynchronize instead). */
AIDAX_API int  aidax_pool_process(aidax_pool* p, const float* in, float* out, uint32_t n_frames);

/* The same pass for a host that streams blocks: submit() stages block k (pinned copy, upload on a copy stream, the
 * pass, download on another copy stream) and returns; collect() waits for the OLDEST submitted block and copies it
 * out. With one block kept in flight — submit(k+1) before collect(k) — the upload of k+1 and the download of k-1 run
 * under the pass of k; with two — submit(k+2) before collect(k) — the host's own round trip (a download, the caller's
 * turn-around, an upload) is off the GPU's critical path as well and the pass is what bounds the rate. At most three
 * blocks between submit and collect (AIDAX_ERR_STATE beyond); collect's n_frames is the submitted block's. One caller
 * thread. The first submit allocates the staging sets: make it before going real-time. */
AIDAX_API int  aidax_pool_submit(aidax_pool* p, const float* in, uint32_t n_frames);
AIDAX_API int  aidax_pool_collect(aidax_pool* p, float* out, uint32_t n_frames);

/* A host that hands over the SAME buffers block after block (the reference's run() gets the host's port buffers,
 * rt-neural-generic.cpp:484-487) can have them pinned once: register_host() page-locks a range of the caller's memory
 * for the pool's device (set-up side: it allocates and may block; the range must stay mapped until unregister_host() or
 * the pool's end). submit() then uploads a block that lies inside a registered range straight out of it — no copy into
 * the pool's staging — and submit_to() names the block's destination up front, so that the download lands there as well
 * and collect() (same `out`) only waits. Buffers outside any registered range take the staged path as before.
 * OWNERSHIP: a block that lies inside a registered range is read by the copy engine AFTER submit() has returned, and a
 * registered `out` is written BEFORE collect() returns — such buffers belong to the pool from submit()/submit_to() until the
 * collect() of that block. With one block kept in flight the host therefore alternates TWO in/out buffer pairs (write block
 * k+1's input while block k's is still the pool's); a host with a single pair collects before it reuses it. (Unregistered
 * buffers are copied into the pool's staging inside submit() and out of it inside collect(): theirs is the usual lifetime.) */
AIDAX_API int  aidax_pool_register_host(aidax_pool* p, void* base, size_t bytes);
AIDAX_API int  aidax_pool_unregister_host(aidax_pool* p, void* base);
AIDAX_API int  aidax_pool_submit_to(aidax_pool* p, const float* in, float* out, uint32_t n_frames);

/* Same pass with device-resident buffers, asynchronous on `hip_stream`
 * (a hipStream_t; NULL = the pool's own stream). No host sync inside. The pool's control pokes (activate,
 * reset_stream, commit_model) run on its own stream; when consecutive operations sit on different streams the
 * pool puts an event edge between them, so program order is kept. A stream handed in here must stay valid
 * until a later call names another one (or NULL), or the pool is destroyed. */
AIDAX_API int  aidax_pool_process_device(aidax_pool* p, const float* d_in, float* d_out,
                                         uint32_t n_frames, void* hip_stream);
/* Waits for the pool's passes. AIDAX_ERR_DEVICE also when a pass since the last report went wrong on the device:
 * the stacked-model kernel k_mfma_lp runs a stream group's layers on separate workgroups that wait for each other, and
 * gives a wait up after 250 ms (another process holding the GPU's CUs); the blocking aidax_pool_process reports the same
 * for its own block and returns silence. The pool serves the model with the one-workgroup-per-group kernel from then on. */
AIDAX_API int  aidax_pool_sync(aidax_pool* p);

/* testModel() (:900-955) on the GPU: input_batch through the bare model from
 * reset state with gains forced to 1 and params forced to 0, compared with
 * output_batch at TEST_MODEL_THR = 1e-5 (rt-neural-generic.h:182). Uses a
 * scratch single-stream pool on `device_id`. out_opt (n_golden floats) may be NULL. */
AIDAX_API int  aidax_model_self_test(const aidax_model* m, int device_id, int32_t* n_errors,
                                     float* max_error, float* out_opt);

/* Bare applyModel (:148-240) from reset state for arbitrary conditioned input:
 * X is [n][input_size] (audio, p1, p2 per sample), y is [n]; gains and skip per
 * the model unless unit_gains != 0. For parity tests against NN-only fixtures. */
AIDAX_API int  aidax_model_forward(const aidax_model* m, int device_id, const float* X, float* y,
                                   uint32_t n, int unit_gains);

/* Introspection for tests: copies one stream's recurrent state (h then c of
 * rnn layer `layer`) to host. Returns hidden size or <0. */
AIDAX_API int  aidax_pool_read_state(aidax_pool* p, uint32_t stream, int layer, float* h, float* c, uint32_t cap);

/* Name of the kernel instantiation a loaded pool dispatches to (profiling aid). */
AIDAX_API const char* aidax_pool_kernel_name(const aidax_pool* p);

/* --------------------------------------------------------------------- hub
 * Host-side stream aggregator (new relative to the reference, whose seam is the
 * per-instance DSP section rt-neural-generic.cpp:621-659). Many plugin instances of ONE process share one
 * pool: each instance attaches to a slot and calls aidax_hub_run() from its run(); the hub launches one
 * pool pass per audio period for everybody.
 *
 * Hosts call the instances of a period one after another on one thread, or in parallel on several; a
 * rendezvous inside run() would deadlock the first kind, so the hub is pipelined by ONE period instead:
 * run() of period p stages the instance's input block and returns the output of period p-1 (silence in the
 * first period). The pass of a period is launched (asynchronously: H2D, kernels, D2H) by the hub's launcher
 * thread as soon as every attached instance has submitted — or when the period's DEADLINE passes (by default
 * half the period after its first submission, aidax_hub_set_deadline_us), so an instance that stalls or
 * stops calling cannot hold the others: they keep their one period of latency, the straggler's stream simply
 * does not advance in that pass. An instance that comes around again before the period was closed, or a change
 * of block size, closes it on the spot. An instance reads the output of the pass that carried ITS previous block
 * while that is at most two passes old and of the same length (a period that was closed in pieces), silence
 * otherwise. run() itself waits only for the event of that pass, outside the hub's lock; the staging buffers
 * rotate over four passes, and a host that closes periods faster than the GPU finishes them waits for the pass
 * four back before its staging is reused. Controls, the loading flag and activate() of an instance whose block still
 * waits for its pass close the period first: a submitted block plays under what was in force when it was submitted. Report aidax_hub_latency_frames() to the host as the plugin's latency. Thread-safe. */
typedef struct aidax_hub aidax_hub;

AIDAX_API int  aidax_hub_create(uint32_t max_instances, uint32_t max_frames, double host_samplerate,
                                int device_id, aidax_hub** out);
AIDAX_API void aidax_hub_destroy(aidax_hub* h);
/* every attached instance plays this model (instances with another model belong to another hub) */
AIDAX_API int  aidax_hub_set_model(aidax_hub* h, const aidax_model* m, int start_mode);
/* a new instance: *slot receives its index; its stream starts from instantiate() + warm-up state */
AIDAX_API int  aidax_hub_attach(aidax_hub* h, int32_t* slot);
/* An instance that plays on (prev, prev_slot) and changes its model file moves to the hub of the new file — a model
 * swap of the reference (work() :807-836, work_response() :859-893) spread over two hubs:
 *   aidax_hub_attach_successor   worker thread: a seat in `h` whose fresh DynamicModel is built (and warmed up) around
 *                                the PARAM targets the predecessor's model holds (:822-825); launches the
 *                                predecessor's pending block and waits for it, outside the hubs' locks. prev may be
 *                                NULL (first load: plain attach).
 *   aidax_hub_adopt              audio thread, at the swap: the new seat takes over the plugin's own DSP members
 *                                (aidax_stream_dsp: biquad memories, gain smoothers) by a device-side copy ordered
 *                                behind the predecessor's last pass. No wait, no allocation. The predecessor's seat
 *                                is detached afterwards (worker), as the old DynamicModel is freed there (:838-840). */
AIDAX_API int  aidax_hub_attach_successor(aidax_hub* h, aidax_hub* prev, int32_t prev_slot, int32_t* slot);
AIDAX_API int  aidax_hub_adopt(aidax_hub* h, int32_t slot, aidax_hub* prev, int32_t prev_slot);
AIDAX_API int  aidax_hub_detach(aidax_hub* h, int32_t slot);
AIDAX_API int  aidax_hub_set_controls(aidax_hub* h, int32_t slot, const aidax_controls* c);
/* the instance's `loading` flag and activate(), as aidax_pool_set_loading / aidax_pool_activate for its stream */
AIDAX_API int  aidax_hub_set_loading(aidax_hub* h, int32_t slot, int loading);
AIDAX_API int  aidax_hub_activate(aidax_hub* h, int32_t slot);
/* the instance's run(): in/out are its n_frames-long port buffers (may alias) */
AIDAX_API int  aidax_hub_run(aidax_hub* h, int32_t slot, const float* in, float* out, uint32_t n_frames);
/* deadline of a period, measured from its first submission: < 0 half the period (default), 0 none
 * (passes are launched only when everybody submitted, on re-entry, or by aidax_hub_flush) */
AIDAX_API int  aidax_hub_set_deadline_us(aidax_hub* h, int64_t microseconds);
/* close the period being collected now (hosts that know their graph is done; tests) */
AIDAX_API int  aidax_hub_flush(aidax_hub* h);
AIDAX_API uint32_t aidax_hub_latency_frames(const aidax_hub* h);
AIDAX_API uint32_t aidax_hub_attached(const aidax_hub* h);
/* the longest block aidax_hub_run accepts (callers slice longer host blocks: the LV2 shell does) */
AIDAX_API uint32_t aidax_hub_max_frames(const aidax_hub* h);
/* number of pool passes launched so far (tests, statistics) */
AIDAX_API uint64_t aidax_hub_launches(const aidax_hub* h);
/* ... of which were launched by the deadline with somebody missing */
AIDAX_API uint64_t aidax_hub_deadline_launches(const aidax_hub* h);
/* Diagnostic: hand-over give-ups of the stacked-model kernel (see aidax_pool_sync) that the hub could attribute to no pass of a chained
 * kernel — nothing was silenced for them; a non-zero count on a healthy system is a bug report. */
AIDAX_API uint64_t aidax_hub_faults_unmapped(const aidax_hub* h);

#ifdef __cplusplus
}
#endif
#endif /* AIDAX_H */
