/*
 * lv2_min.h — the slice of the LV2 C ABI this plugin shell uses, restated from
 * the public LV2 specification (lv2plug.in; core, urid, atom, worker, state,
 * log, patch). The LV2 headers are not installed in this image, and the ABI is
 * stable C, so the struct layouts and URIs below are written out here; where a
 * system <lv2/...> is available these declarations are layout-identical.
 *
 * The reference includes the real headers at
 * rt-neural-generic/src/rt-neural-generic.h:33-41 and uris.h:21-24.
 */
#ifndef AIDAX_LV2_MIN_H
#define AIDAX_LV2_MIN_H

#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ core */
#define LV2_CORE_URI "http://lv2plug.in/ns/lv2core"
typedef void* LV2_Handle;

typedef struct {
    const char* URI;
    void*       data;
} LV2_Feature;

typedef struct LV2_Descriptor {
    const char* URI;
    LV2_Handle (*instantiate)(const struct LV2_Descriptor* descriptor, double sample_rate,
                              const char* bundle_path, const LV2_Feature* const* features);
    void (*connect_port)(LV2_Handle instance, uint32_t port, void* data_location);
    void (*activate)(LV2_Handle instance);
    void (*run)(LV2_Handle instance, uint32_t sample_count);
    void (*deactivate)(LV2_Handle instance);
    void (*cleanup)(LV2_Handle instance);
    const void* (*extension_data)(const char* uri);
} LV2_Descriptor;

#define LV2_SYMBOL_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ urid */
#define LV2_URID__map "http://lv2plug.in/ns/ext/urid#map"
typedef uint32_t LV2_URID;
typedef void*    LV2_URID_Map_Handle;
typedef struct {
    LV2_URID_Map_Handle handle;
    LV2_URID (*map)(LV2_URID_Map_Handle handle, const char* uri);
} LV2_URID_Map;

/* ------------------------------------------------------------------ atom */
#define LV2_ATOM_URI "http://lv2plug.in/ns/ext/atom"
#define LV2_ATOM__Blank         LV2_ATOM_URI "#Blank"
#define LV2_ATOM__Float         LV2_ATOM_URI "#Float"
#define LV2_ATOM__Object        LV2_ATOM_URI "#Object"
#define LV2_ATOM__Path          LV2_ATOM_URI "#Path"
#define LV2_ATOM__Resource      LV2_ATOM_URI "#Resource"
#define LV2_ATOM__Sequence      LV2_ATOM_URI "#Sequence"
#define LV2_ATOM__URID          LV2_ATOM_URI "#URID"
#define LV2_ATOM__eventTransfer LV2_ATOM_URI "#eventTransfer"

typedef struct { uint32_t size; uint32_t type; } LV2_Atom;
typedef struct { LV2_Atom atom; uint32_t body; } LV2_Atom_URID;
typedef struct { uint32_t unit; uint32_t pad; } LV2_Atom_Sequence_Body;
typedef struct { LV2_Atom atom; LV2_Atom_Sequence_Body body; } LV2_Atom_Sequence;
typedef struct {
    union { int64_t frames; double beats; } time;
    LV2_Atom body;
} LV2_Atom_Event;
typedef struct { uint32_t id; uint32_t otype; } LV2_Atom_Object_Body;
typedef struct { LV2_Atom atom; LV2_Atom_Object_Body body; } LV2_Atom_Object;
typedef struct { uint32_t key; uint32_t context; LV2_Atom value; } LV2_Atom_Property_Body;

static inline uint32_t lv2_atom_pad_size(uint32_t size) { return (size + 7U) & (~7U); }

/* ---------------------------------------------------------------- worker */
#define LV2_WORKER_URI "http://lv2plug.in/ns/ext/worker"
#define LV2_WORKER__interface LV2_WORKER_URI "#interface"
#define LV2_WORKER__schedule  LV2_WORKER_URI "#schedule"

typedef enum {
    LV2_WORKER_SUCCESS = 0,
    LV2_WORKER_ERR_UNKNOWN = 1,
    LV2_WORKER_ERR_NO_SPACE = 2
} LV2_Worker_Status;

typedef void* LV2_Worker_Respond_Handle;
typedef LV2_Worker_Status (*LV2_Worker_Respond_Function)(LV2_Worker_Respond_Handle handle, uint32_t size, const void* data);

typedef struct {
    LV2_Worker_Status (*work)(LV2_Handle instance, LV2_Worker_Respond_Function respond,
                              LV2_Worker_Respond_Handle handle, uint32_t size, const void* data);
    LV2_Worker_Status (*work_response)(LV2_Handle instance, uint32_t size, const void* body);
    LV2_Worker_Status (*end_run)(LV2_Handle instance);
} LV2_Worker_Interface;

typedef void* LV2_Worker_Schedule_Handle;
typedef struct {
    LV2_Worker_Schedule_Handle handle;
    LV2_Worker_Status (*schedule_work)(LV2_Worker_Schedule_Handle handle, uint32_t size, const void* data);
} LV2_Worker_Schedule;

/* ----------------------------------------------------------------- state */
#define LV2_STATE_URI "http://lv2plug.in/ns/ext/state"
#define LV2_STATE__interface LV2_STATE_URI "#interface"
#define LV2_STATE__mapPath   LV2_STATE_URI "#mapPath"
#define LV2_STATE__freePath  LV2_STATE_URI "#freePath"

typedef void* LV2_State_Handle;
typedef void* LV2_State_Map_Path_Handle;
typedef void* LV2_State_Free_Path_Handle;

typedef enum { LV2_STATE_IS_POD = 1, LV2_STATE_IS_PORTABLE = 1 << 1, LV2_STATE_IS_NATIVE = 1 << 2 } LV2_State_Flags;
typedef enum {
    LV2_STATE_SUCCESS = 0, LV2_STATE_ERR_UNKNOWN = 1, LV2_STATE_ERR_BAD_TYPE = 2, LV2_STATE_ERR_BAD_FLAGS = 3,
    LV2_STATE_ERR_NO_FEATURE = 4, LV2_STATE_ERR_NO_PROPERTY = 5, LV2_STATE_ERR_NO_SPACE = 6
} LV2_State_Status;

typedef LV2_State_Status (*LV2_State_Store_Function)(LV2_State_Handle handle, uint32_t key, const void* value,
                                                     size_t size, uint32_t type, uint32_t flags);
typedef const void* (*LV2_State_Retrieve_Function)(LV2_State_Handle handle, uint32_t key, size_t* size,
                                                   uint32_t* type, uint32_t* flags);
typedef struct {
    LV2_State_Status (*save)(LV2_Handle instance, LV2_State_Store_Function store, LV2_State_Handle handle,
                             uint32_t flags, const LV2_Feature* const* features);
    LV2_State_Status (*restore)(LV2_Handle instance, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                uint32_t flags, const LV2_Feature* const* features);
} LV2_State_Interface;

typedef struct {
    LV2_State_Map_Path_Handle handle;
    char* (*abstract_path)(LV2_State_Map_Path_Handle handle, const char* absolute_path);
    char* (*absolute_path)(LV2_State_Map_Path_Handle handle, const char* abstract_path);
} LV2_State_Map_Path;

typedef struct {
    LV2_State_Free_Path_Handle handle;
    void (*free_path)(LV2_State_Free_Path_Handle handle, char* path);
} LV2_State_Free_Path;

/* ------------------------------------------------------------------- log */
#define LV2_LOG_URI "http://lv2plug.in/ns/ext/log"
#define LV2_LOG__log     LV2_LOG_URI "#log"
#define LV2_LOG__Error   LV2_LOG_URI "#Error"
#define LV2_LOG__Note    LV2_LOG_URI "#Note"
#define LV2_LOG__Trace   LV2_LOG_URI "#Trace"
#define LV2_LOG__Warning LV2_LOG_URI "#Warning"

typedef void* LV2_Log_Handle;
typedef struct {
    LV2_Log_Handle handle;
    int (*printf)(LV2_Log_Handle handle, LV2_URID type, const char* fmt, ...);
    int (*vprintf)(LV2_Log_Handle handle, LV2_URID type, const char* fmt, va_list ap);
} LV2_Log_Log;

/* --------------------------------------------------- patch / midi / params */
#define LV2_PATCH_URI "http://lv2plug.in/ns/ext/patch"
#define LV2_PATCH__Get      LV2_PATCH_URI "#Get"
#define LV2_PATCH__Set      LV2_PATCH_URI "#Set"
#define LV2_PATCH__property LV2_PATCH_URI "#property"
#define LV2_PATCH__value    LV2_PATCH_URI "#value"
#define LV2_MIDI__MidiEvent "http://lv2plug.in/ns/ext/midi#MidiEvent"
#define LV2_PARAMETERS__gain "http://lv2plug.in/ns/ext/parameters#gain"

#ifdef __cplusplus
}
#endif
#endif
