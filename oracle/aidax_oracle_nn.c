/*
 * aidax_oracle_nn.c — CPU oracle, NN half + multi-thread CPU baseline timer.
 * TEST INFRASTRUCTURE ONLY (see aidax_oracle.h). Floating point here is
 * compared to tolerance (1e-5, rt-neural-generic.h:182), never bit-for-bit,
 * so this unit may be built with FMA contraction.
 */
#include "aidax_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* per-stream state on its own cache lines: threads of the CPU baseline each own a
 * contiguous stream range and must not false-share with their neighbours */
void* orc_calloc_lines(size_t bytes)
{
    void* p = NULL;
    const size_t sz = (bytes + 127) & ~(size_t)127;
    if (posix_memalign(&p, 128, sz ? sz : 128) != 0) return NULL;
    memset(p, 0, sz ? sz : 128);
    return p;
}

#define REAL float
#define SUFFIX _f32
#define RE_EXP expf
#define RE_TANH tanhf
#include "nn_impl.inc"
#undef REAL
#undef SUFFIX
#undef RE_EXP
#undef RE_TANH

#define REAL double
#define SUFFIX _f64
#define RE_EXP exp
#define RE_TANH tanh
#include "nn_impl.inc"
#undef REAL
#undef SUFFIX
#undef RE_EXP
#undef RE_TANH

/* Third flavour, used only by the CPU-baseline timer: fp32 with branch-free exp / tanh that gcc
 * vectorises (AVX2), standing in for the xsimd kernels of the reference's RTNEURAL_XSIMD build
 * (rt-neural-generic/CMakeLists.txt:10) so that the baseline is not handicapped by scalar libm calls.
 * exp: round-to-nearest range reduction + degree-6 polynomial (<= 2 ulp on [-87,87]);
 * tanh: the odd rational also used on the GPU (<= 3.6e-7 relative). Checked against the goldens
 * in tests/test_oracle.py like the libm flavour. */
static inline float vexpf(float x)
{
    x = x < -87.0f ? -87.0f : (x > 87.0f ? 87.0f : x);
    const float n = __builtin_roundevenf(x * 1.44269504088896341f);
    float r = x - n * 0.693359375f;             /* ln2 split hi/lo */
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * r * r + r + 1.0f;
    union { float f; int i; } u;
    u.i = ((int)n + 127) << 23;
    return p * u.f;
}
static inline float vtanhf(float v)
{
    const float x = v < -7.9f ? -7.9f : (v > 7.9f ? 7.9f : v);
    const float u = x * x;
    float p = -8.488730763828322e-14f;
    p = p * u + 5.277955823366522e-11f;
    p = p * u + -2.0225239996482085e-08f;
    p = p * u + 1.1154311501654368e-05f;
    p = p * u + 0.003103956503888039f;
    p = p * u + 0.13084010352004496f;
    p = p * u + 0.9999999933888696f;
    float q = 0.00025461456545097517f;
    q = q * u + 0.02449517952619233f;
    q = q * u + 0.46417337453820245f;
    q = q * u + 1.0f;
    return (p * x) / q;
}
#define REAL float
#define SUFFIX _fast
#define RE_EXP vexpf
#define RE_TANH vtanhf
#include "nn_impl.inc"
#undef REAL
#undef SUFFIX
#undef RE_EXP
#undef RE_TANH

struct orc_net { int f64; int in_size; net_f32* a; net_f64* b; net_fast* c; };

orc_net* orc_net_create(const orc_layer_desc* layers, int n_layers, int use_f64)
{
    orc_net* n = (orc_net*)orc_calloc_lines(sizeof(*n));
    n->f64 = use_f64;
    n->in_size = layers[0].in_size;
    if (use_f64 == 1) n->b = net_create_f64(layers, n_layers);
    else if (use_f64 == 2) n->c = net_create_fast(layers, n_layers);
    else n->a = net_create_f32(layers, n_layers);
    return n;
}

void orc_net_free(orc_net* n)
{
    if (!n) return;
    if (n->a) net_free_f32(n->a);
    if (n->b) net_free_f64(n->b);
    if (n->c) net_free_fast(n->c);
    free(n);
}

void orc_net_reset(orc_net* n) { if (n->f64 == 1) net_reset_f64(n->b); else if (n->f64 == 2) net_reset_fast(n->c); else net_reset_f32(n->a); }
int orc_net_in_size(const orc_net* n) { return n->in_size; }
float orc_net_forward(orc_net* n, const float* x)
{
    return n->f64 == 1 ? net_forward_f64(n->b, x) : n->f64 == 2 ? net_forward_fast(n->c, x) : net_forward_f32(n->a, x);
}

int orc_net_state(const orc_net* n, int layer, float* h, float* c, int cap)
{
    int H;
    if (n->f64 == 2) {
        if (layer >= n->c->n_layers) return -1;
        layer_fast* L = &n->c->layers[layer]; H = L->out_size;
        for (int j = 0; j < H && j < cap; ++j) { h[j] = L->h[j]; c[j] = L->c[j]; }
    } else if (n->f64) {
        if (layer >= n->b->n_layers) return -1;
        layer_f64* L = &n->b->layers[layer]; H = L->out_size;
        for (int j = 0; j < H && j < cap; ++j) { h[j] = (float)L->h[j]; c[j] = (float)L->c[j]; }
    } else {
        if (layer >= n->a->n_layers) return -1;
        layer_f32* L = &n->a->layers[layer]; H = L->out_size;
        for (int j = 0; j < H && j < cap; ++j) { h[j] = L->h[j]; c[j] = L->c[j]; }
    }
    return H;
}

/* ------------------------------------------------------------ CPU baseline */

typedef struct {
    orc_plugin** plugins; const orc_controls* c;
    int s0, s1, n_frames, n_blocks;
    const float* in; float* out;
    pthread_barrier_t* bar;
} bench_arg;

static void* bench_thread(void* p)
{
    bench_arg* a = (bench_arg*)p;
    pthread_barrier_wait(a->bar);
    for (int b = 0; b < a->n_blocks; ++b)
        for (int s = a->s0; s < a->s1; ++s)
            orc_plugin_run(a->plugins[s], a->c, a->in + (size_t)s * a->n_frames,
                           a->out + (size_t)s * a->n_frames, (uint32_t)a->n_frames);
    pthread_barrier_wait(a->bar);
    return NULL;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_bench(const orc_layer_desc* layers, int n_layers, int input_size, int input_skip,
                 float in_gain, float out_gain, const orc_controls* c,
                 int n_streams, int n_frames, int n_blocks, int warm_blocks, int n_threads,
                 const float* in, float* out_last, int flavour)
{
    /* plugin structs padded to whole cache lines for the same reason */
    const size_t pstride = (sizeof(orc_plugin) + 127) & ~(size_t)127;
    char* pmem = (char*)orc_calloc_lines(pstride * (size_t)n_streams);
    orc_plugin** plugins = (orc_plugin**)calloc((size_t)n_streams, sizeof(orc_plugin*));
    for (int s = 0; s < n_streams; ++s) plugins[s] = (orc_plugin*)(pmem + pstride * (size_t)s);
    float* out = (float*)calloc((size_t)n_streams * (size_t)n_frames, sizeof(float));
    for (int s = 0; s < n_streams; ++s) {
        orc_plugin_init(plugins[s], 48000.0);
        orc_net* net = orc_net_create(layers, n_layers, flavour);
        orc_dynmodel* m = orc_dynmodel_create(net, input_size, input_skip, in_gain, out_gain,
                                              48000.0f, 0.f, 0.f, 1);
        orc_plugin_set_model(plugins[s], m);
    }
    /* untimed warm-up blocks, single pass on the calling thread pool below */
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_streams) n_threads = n_streams;
    double elapsed = 0.0;
    for (int phase = 0; phase < 2; ++phase) {
        const int blocks = phase == 0 ? warm_blocks : n_blocks;
        if (blocks <= 0) continue;
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, NULL, (unsigned)n_threads + 1);
        pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
        bench_arg* args = (bench_arg*)calloc((size_t)n_threads, sizeof(bench_arg));
        for (int t = 0; t < n_threads; ++t) {
            args[t].plugins = plugins; args[t].c = c;
            args[t].s0 = (int)((long long)n_streams * t / n_threads);
            args[t].s1 = (int)((long long)n_streams * (t + 1) / n_threads);
            args[t].n_frames = n_frames; args[t].n_blocks = blocks;
            args[t].in = in; args[t].out = out; args[t].bar = &bar;
            pthread_create(&th[t], NULL, bench_thread, &args[t]);
        }
        pthread_barrier_wait(&bar);
        const double t0 = now_s();
        pthread_barrier_wait(&bar);
        const double t1 = now_s();
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&bar);
        free(th); free(args);
        if (phase == 1) elapsed = t1 - t0;
    }
    if (out_last) memcpy(out_last, out, sizeof(float) * (size_t)n_streams * (size_t)n_frames);
    for (int s = 0; s < n_streams; ++s) orc_dynmodel_free(plugins[s]->model);
    free(plugins); free(pmem); free(out);
    return elapsed;
}
