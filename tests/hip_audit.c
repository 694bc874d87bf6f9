/* hip_audit.c — LD_PRELOAD interposer that counts the HIP runtime calls a thread makes between
 * hip_audit_begin() and hip_audit_end(). Test infrastructure (tests/test_rt_contract.py): it proves from
 * OUTSIDE the library that the audio-thread entry points (work_response(), run(), aidax_pool_commit_model,
 * aidax_hub_run) neither allocate nor free nor wait for the device — the contract the reference keeps by
 * doing all of that in work() (rt-neural-generic.cpp:807-893).
 *
 * Built by the test with plain gcc; no HIP headers needed (handles are opaque pointers, hipError_t is int). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

enum {
    A_MALLOC, A_FREE, A_HOST_MALLOC, A_HOST_FREE, A_DEVICE_SYNC, A_STREAM_SYNC, A_EVENT_SYNC, A_MEMCPY_SYNC,
    A_STREAM_CREATE, A_EVENT_CREATE, A_MEMCPY_ASYNC, A_LAUNCH, A_EVENT_RECORD, A_STREAM_WAIT_EVENT, A_STREAM_WRITE_VALUE, A_COUNT
};
static const char* const kNames[A_COUNT] = {
    "hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipDeviceSynchronize", "hipStreamSynchronize",
    "hipEventSynchronize", "hipMemcpy", "hipStreamCreate", "hipEventCreate", "hipMemcpyAsync", "launch",
    "hipEventRecord", "hipStreamWaitEvent", "hipStreamWriteValue32"
};

static __thread int t_on = 0;
static __thread uint64_t t_cnt[A_COUNT];

void hip_audit_begin(void) { memset(t_cnt, 0, sizeof t_cnt); t_on = 1; }
void hip_audit_end(uint64_t* out) { t_on = 0; if (out) memcpy(out, t_cnt, sizeof t_cnt); }
int hip_audit_fields(void) { return A_COUNT; }
const char* hip_audit_name(int i) { return (i >= 0 && i < A_COUNT) ? kNames[i] : ""; }

static void* real(const char* name)
{
    void* f = dlsym(RTLD_NEXT, name);
    if (!f) {
        /* the runtime was dlopen'ed RTLD_LOCAL by somebody (python extension modules): ask it by soname */
        static const char* const libs[] = { "libamdhip64.so.7", "libamdhip64.so", NULL };
        for (int i = 0; !f && libs[i]; ++i) {
            void* h = dlopen(libs[i], RTLD_LAZY | RTLD_NOLOAD);
            if (!h) h = dlopen(libs[i], RTLD_LAZY);
            if (h) f = dlsym(h, name);
        }
    }
    if (!f) fprintf(stderr, "hip_audit: cannot resolve %s\n", name);
    return f;
}
#define COUNT(i) do { if (t_on) ++t_cnt[i]; } while (0)
#define REAL(name, ...) typedef int (*fn_t)(__VA_ARGS__); static fn_t fn = NULL; if (!fn) fn = (fn_t)real(name)

typedef struct { unsigned x, y, z; } dim3_t;

int hipMalloc(void** p, size_t n) { REAL("hipMalloc", void**, size_t); COUNT(A_MALLOC); return fn(p, n); }
int hipFree(void* p) { REAL("hipFree", void*); COUNT(A_FREE); return fn(p); }
int hipHostMalloc(void** p, size_t n, unsigned f) { REAL("hipHostMalloc", void**, size_t, unsigned); COUNT(A_HOST_MALLOC); return fn(p, n, f); }
int hipHostFree(void* p) { REAL("hipHostFree", void*); COUNT(A_HOST_FREE); return fn(p); }
int hipDeviceSynchronize(void) { REAL("hipDeviceSynchronize", void); COUNT(A_DEVICE_SYNC); return fn(); }
int hipStreamSynchronize(void* s) { REAL("hipStreamSynchronize", void*); COUNT(A_STREAM_SYNC); return fn(s); }
int hipEventSynchronize(void* e) { REAL("hipEventSynchronize", void*); COUNT(A_EVENT_SYNC); return fn(e); }
int hipMemcpy(void* d, const void* s, size_t n, int k) { REAL("hipMemcpy", void*, const void*, size_t, int); COUNT(A_MEMCPY_SYNC); return fn(d, s, n, k); }
int hipStreamCreateWithFlags(void** s, unsigned f) { REAL("hipStreamCreateWithFlags", void**, unsigned); COUNT(A_STREAM_CREATE); return fn(s, f); }
int hipEventCreateWithFlags(void** e, unsigned f) { REAL("hipEventCreateWithFlags", void**, unsigned); COUNT(A_EVENT_CREATE); return fn(e, f); }
int hipMemcpyAsync(void* d, const void* s, size_t n, int k, void* q)
{
    REAL("hipMemcpyAsync", void*, const void*, size_t, int, void*);
    COUNT(A_MEMCPY_ASYNC);
    return fn(d, s, n, k, q);
}
int hipLaunchKernel(const void* f, dim3_t g, dim3_t b, void** args, size_t shmem, void* q)
{
    REAL("hipLaunchKernel", const void*, dim3_t, dim3_t, void**, size_t, void*);
    COUNT(A_LAUNCH);
    return fn(f, g, b, args, shmem, q);
}
int hipExtLaunchKernel(const void* f, dim3_t g, dim3_t b, void** args, size_t shmem, void* q, void* start, void* stop, int flags)
{
    /* (the pipelined host path: the pass's "done" event rides on its dispatch) */
    REAL("hipExtLaunchKernel", const void*, dim3_t, dim3_t, void**, size_t, void*, void*, void*, int);
    COUNT(A_LAUNCH);
    return fn(f, g, b, args, shmem, q, start, stop, flags);
}
int hipEventRecord(void* e, void* q) { REAL("hipEventRecord", void*, void*); COUNT(A_EVENT_RECORD); return fn(e, q); }
int hipStreamWaitEvent(void* q, void* e, unsigned f) { REAL("hipStreamWaitEvent", void*, void*, unsigned); COUNT(A_STREAM_WAIT_EVENT); return fn(q, e, f); }
int hipStreamWriteValue32(void* q, void* p, uint32_t v, unsigned f) { REAL("hipStreamWriteValue32", void*, void*, uint32_t, unsigned); COUNT(A_STREAM_WRITE_VALUE); return fn(q, p, v, f); }
