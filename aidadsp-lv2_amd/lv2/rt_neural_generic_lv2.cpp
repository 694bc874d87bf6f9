// rt_neural_generic_lv2.cpp — the LV2 plugin shell of AIDA-X rt-neural-generic on top of the
// C ABI of include/aidax.h. Builds rt-neural-generic.so exporting lv2_descriptor().
//
// It keeps the reference plugin's host-visible behaviour (generic build: AIDADSP_COMMERCIAL=0,
// AIDADSP_MODEL_LOADER=1, AIDADSP_CONDITIONED_MODELS=1, two params, optional DC blocker):
//   descriptor / URI                    rt-neural-generic/src/rt-neural-generic.cpp:11-29, uris.h:26-31
//   25 ports, same indices              rt-neural-generic.h:84-112, rt-neural-generic.ttl:61-313
//   instantiate / features              rt-neural-generic.cpp:244-333
//   patch:Set{#json, atom:Path} on CONTROL -> worker load       :524-586
//   worker protocol load/apply/free     :807-893, message structs rt-neural-generic.h:132-157
//   NOTIFY echo of the applied file     :880-887, uris.h:76-94
//   state save/restore of the model path :707-796
//   mute while loading, ModelInSize port :518, :654
// All DSP (run() :621-659) happens on the GPU behind aidax_pool_process(); each plugin instance
// is one stream of a one-stream pool. There is no CPU fallback: without a usable HIP device
// instantiate() returns NULL, like the reference does for a missing host feature (:265-273).
//
// Threads, as in the reference: work() (worker thread) loads the json AND prepares the model on the GPU —
// weight upload, state reset, 2048-frame warm-up, every allocation (aidax_pool_prepare_model, or the hub seat
// in hub mode); work_response() (audio thread) only swaps handles (aidax_pool_commit_model: no allocation, no
// free, no wait) and hands the old model back to the worker for deletion (kWorkerFree). run() waits for nothing
// but the stream that carries its own block.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aidax.h"
#include "lv2_min.h"

#define PLUGIN_URI "http://aidadsp.cc/plugins/aidadsp-bundle/rt-neural-generic"
#define PLUGIN__json PLUGIN_URI "#json"
#define PLUGIN__applyJson PLUGIN_URI "#applyJson"

namespace {

// LATENCY (index 25) is not in the reference: the lv2:reportsLatency output (one period in hub mode / else 0).
// Hosts that follow the reference's TTL never connect it.
enum PortIndex {   // ports_t, rt-neural-generic.h:84-112 (generic build)
    IN = 0, OUT_1, PLUGIN_CONTROL, PLUGIN_NOTIFY, IN_LPF, PREGAIN, NET_BYPASS, PARAM1, PARAM2,
    EQ_BYPASS, EQ_POS, BASS, BFREQ, MID, MFREQ, MIDQ, MTYPE, TREBLE, TFREQ, DEPTH, PRESENCE,
    DCBLOCKER, MASTER, INPUT_SIZE, PLUGIN_ENABLED, LATENCY, PLUGIN_PORT_COUNT
};

enum WorkerMessageType { kWorkerLoad, kWorkerApply, kWorkerFree, kWorkerNote };   // rt-neural-generic.h:132-136 (+ kWorkerNote: a log line the audio thread wants said)
struct WorkerMessage { WorkerMessageType type; };
struct WorkerLoadMessage { WorkerMessageType type; char path[1024]; };      // :144-151
// :154-157 with the C-ABI handles: the model, what aidax_pool_prepare_model staged for it (after the swap: what the
// swap retired), and in hub mode the seat (hub, slot) the worker attached for this instance
struct WorkerNoteMessage { WorkerMessageType type; uint32_t n_samples, slices, slice_len, cap; };
struct WorkerApplyMessage { WorkerMessageType type; aidax_model* model; aidax_staged* staged; aidax_hub* hub; int32_t slot; };

struct PluginURIs {   // uris.h:33-48
    LV2_URID atom_Float, atom_Path, atom_Resource, atom_Sequence, atom_URID, atom_eventTransfer;
    LV2_URID atom_Object, atom_Blank;
    LV2_URID applyJson, json, midi_Event, param_gain, patch_Get, patch_Set, patch_property, patch_value;
    LV2_URID log_Error, log_Note, log_Trace;
};

struct Plugin {
    // ports
    const float* in = nullptr;
    float* out_1 = nullptr;
    const LV2_Atom_Sequence* control_port = nullptr;
    LV2_Atom_Sequence* notify_port = nullptr;
    const float* ctl[PLUGIN_PORT_COUNT] = {};
    float* input_size = nullptr;
    float* latency = nullptr;

    // features
    LV2_URID_Map* map = nullptr;
    LV2_Worker_Schedule* schedule = nullptr;
    LV2_Log_Log* log = nullptr;
    PluginURIs uris{};

    double samplerate = 48000.0;
    bool loading = true;
    int last_input_size = 0;
    aidax_pool* pool = nullptr;      // this instance's own one-stream pool (default mode)
    aidax_model* model = nullptr;
    // hub mode (AIDAX_HUB=<instances per model>): instances that play the same model file share one pool
    // pass per audio period through aidax_hub, at one period of latency
    int hub_capacity = 0;
    int device = 0;
    bool strict = true;              // only the reference's 54 architectures (AIDAX_STRICT_REFERENCE_SET=0 lifts it)
    uint32_t max_frames = 8192;      // of this instance's pool: longer host blocks are processed in chunks
    uint32_t error_count = 0;
    uint32_t sliced_n = 0, slice_len = 0;   // hub mode: the slice length chosen for host blocks of sliced_n frames (run())
    aidax_hub* hub = nullptr;
    int32_t slot = -1;

    aidax_controls last_controls{};
    bool have_last_controls = false;
    bool last_loading = true;

    // NOTIFY writer (the subset of LV2_Atom_Forge the reference uses)
    uint8_t* notify_buf = nullptr;
    uint32_t notify_capacity = 0;
    bool notify_open = false;
};

// ---- device placement: AIDAX_DEVICE names the candidates ("auto", "0-3,6"; unset = device 0, as before);
// one load count per device and process — instances in one-stream mode, hub seats in hub mode — and aidax_pick_device's rule:
// least loaded, lowest index first. Instances share nothing (rt-neural-generic.h:198-239), so this is all there is to a node
// with several GPUs.
std::mutex g_dev_mu;
uint32_t g_dev_load[256] = {};
int place_on_device(uint32_t weight)
{
    int count = 0;
    if (aidax_device_count(&count) != AIDAX_OK) return -1;
    std::lock_guard<std::mutex> g(g_dev_mu);
    int device = -1;
    if (aidax_pick_device(std::getenv("AIDAX_DEVICE"), count, g_dev_load, &device) != AIDAX_OK) return -1;
    g_dev_load[device] += weight;
    return device;
}
void leave_device(int device, uint32_t weight)
{
    if (device < 0 || device >= 256) return;
    std::lock_guard<std::mutex> g(g_dev_mu);
    g_dev_load[device] = g_dev_load[device] >= weight ? g_dev_load[device] - weight : 0u;
}

// ---- hub registry: the hubs of a model file (one per device that serves it), shared by the instances of this process
struct HubRef { aidax_hub* hub; int refs; int device; };
std::mutex g_hub_mu;
std::multimap<std::string, HubRef> g_hubs;

// worker / main thread: give a seat back; the last one out destroys the hub
void hub_leave(aidax_hub* hub, int32_t slot)
{
    if (!hub) return;
    std::lock_guard<std::mutex> g(g_hub_mu);
    aidax_hub_detach(hub, slot);
    for (auto it = g_hubs.begin(); it != g_hubs.end(); ++it) {
        if (it->second.hub != hub) continue;
        leave_device(it->second.device, 1);
        if (--it->second.refs == 0) {
            aidax_hub_destroy(hub);
            g_hubs.erase(it);
        }
        return;
    }
}

// worker thread: a seat in the hub of `model`'s file, creating the hub (weights upload + warm-up) for the first instance
bool hub_join(Plugin* self, const aidax_model* model, aidax_hub** hub_out, int32_t* slot_out)
{
    const std::string key = aidax_model_path(model);
    std::lock_guard<std::mutex> g(g_hub_mu);
    // Which hub (aidax_pick_hub, a pure rule tested with injected devices): an instance that already plays stays on ITS device — the
    // first hub of this file there with a free seat (the one it sits in now counts: its seat is given back after the swap), else a new
    // hub there — because work_response() carries its biquad memories and gain smoothers into the new seat with a device-side copy
    // (aidax_hub_adopt; the reference keeps them across a swap, rt-neural-generic.cpp:868-875). A first join takes any hub of the file
    // with a free seat, else opens one on the least-loaded device AIDAX_DEVICE allows — how the instances of one file spread over GPUs.
    int current_device = -1;
    if (self->hub)
        for (const auto& kv : g_hubs)
            if (kv.second.hub == self->hub) { current_device = kv.second.device; break; }
    auto range = g_hubs.equal_range(key);
    std::vector<decltype(g_hubs)::iterator> cand;
    std::vector<int> cand_dev;
    std::vector<uint32_t> cand_free;
    for (auto k = range.first; k != range.second; ++k) {
        cand.push_back(k);
        cand_dev.push_back(k->second.device);
        cand_free.push_back(k->second.hub == self->hub ? 1u : k->second.refs < self->hub_capacity ? static_cast<uint32_t>(self->hub_capacity - k->second.refs) : 0u);
    }
    int count = 0, index = -1, device = -1;
    if (aidax_device_count(&count) != AIDAX_OK) return false;
    {
        std::lock_guard<std::mutex> gd(g_dev_mu);
        if (aidax_pick_hub(cand_dev.data(), cand_free.data(), static_cast<int>(cand.size()), current_device, std::getenv("AIDAX_DEVICE"), count,
                           g_dev_load, &index, &device) != AIDAX_OK)
            return false;
    }
    auto it = index >= 0 ? cand[static_cast<size_t>(index)] : g_hubs.end();
    if (it == g_hubs.end()) {
        aidax_hub* hub = nullptr;
        const char* fr = std::getenv("AIDAX_HUB_FRAMES");
        const uint32_t max_frames = fr ? static_cast<uint32_t>(std::atoi(fr)) : 2048u;
        // twice the seats: an instance that reloads a file of this hub holds its old seat until the swap and the worker's
        // kWorkerFree, so all of them may sit here twice for a moment
        if (aidax_hub_create(2u * static_cast<uint32_t>(self->hub_capacity), max_frames, self->samplerate, device, &hub) != AIDAX_OK)
            return false;
        if (const char* dl = std::getenv("AIDAX_HUB_DEADLINE_US")) aidax_hub_set_deadline_us(hub, std::atoll(dl));
        if (aidax_hub_set_model(hub, model, AIDAX_START_WARMUP) != AIDAX_OK) { aidax_hub_destroy(hub); return false; }
        it = g_hubs.emplace(key, HubRef{ hub, 0, device });
    }
    int32_t slot = -1;
    // the seat continues the stream of the one this instance plays on now (PARAM targets for the new DynamicModel, :822-825)
    if (aidax_hub_attach_successor(it->second.hub, self->hub, self->slot, &slot) != AIDAX_OK) {
        if (it->second.refs == 0) { aidax_hub_destroy(it->second.hub); g_hubs.erase(it); }
        return false;
    }
    ++it->second.refs;
    {
        std::lock_guard<std::mutex> gd(g_dev_mu);
        if (it->second.device >= 0 && it->second.device < 256) ++g_dev_load[it->second.device];
    }
    *hub_out = it->second.hub;
    *slot_out = slot;
    return true;
}

void plog(Plugin* self, LV2_URID type, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    if (self->log && self->log->vprintf) self->log->vprintf(self->log->handle, type, fmt, ap);
    else vfprintf(stderr, fmt, ap);
    va_end(ap);
}

void map_plugin_uris(LV2_URID_Map* map, PluginURIs* u)
{
    auto m = [map](const char* uri) { return map->map(map->handle, uri); };
    u->atom_Float = m(LV2_ATOM__Float);       u->atom_Path = m(LV2_ATOM__Path);
    u->atom_Resource = m(LV2_ATOM__Resource); u->atom_Sequence = m(LV2_ATOM__Sequence);
    u->atom_URID = m(LV2_ATOM__URID);         u->atom_eventTransfer = m(LV2_ATOM__eventTransfer);
    u->atom_Object = m(LV2_ATOM__Object);     u->atom_Blank = m(LV2_ATOM__Blank);
    u->applyJson = m(PLUGIN__applyJson);      u->json = m(PLUGIN__json);
    u->midi_Event = m(LV2_MIDI__MidiEvent);   u->param_gain = m(LV2_PARAMETERS__gain);
    u->patch_Get = m(LV2_PATCH__Get);         u->patch_Set = m(LV2_PATCH__Set);
    u->patch_property = m(LV2_PATCH__property); u->patch_value = m(LV2_PATCH__value);
    u->log_Error = m(LV2_LOG__Error);         u->log_Note = m(LV2_LOG__Note);   u->log_Trace = m(LV2_LOG__Trace);
}

// ---- NOTIFY sequence writer: sequence head in run(), one patch:Set event in work_response()
void notify_begin(Plugin* self)
{
    self->notify_open = false;
    if (!self->notify_port) return;
    self->notify_capacity = self->notify_port->atom.size;      // host stores the buffer capacity here (:528)
    self->notify_buf = reinterpret_cast<uint8_t*>(self->notify_port);
    if (self->notify_capacity < sizeof(LV2_Atom_Sequence_Body)) return;
    self->notify_port->atom.type = self->uris.atom_Sequence;
    self->notify_port->atom.size = sizeof(LV2_Atom_Sequence_Body);
    self->notify_port->body.unit = 0;
    self->notify_port->body.pad = 0;
    self->notify_open = true;
}

// write_set_file (uris.h:76-94): [] a patch:Set ; patch:property <#json> ; patch:value </path> .
bool notify_set_file(Plugin* self, const char* filename)
{
    if (!self->notify_open) return false;
    const uint32_t len = static_cast<uint32_t>(std::strlen(filename));
    const uint32_t path_size = len + 2;                         // forge_path(len + 1) stores len + 2 bytes (both NULs)
    const uint32_t prop1 = sizeof(LV2_Atom_Property_Body) + lv2_atom_pad_size(sizeof(uint32_t));      // key + URID atom
    const uint32_t prop2 = sizeof(LV2_Atom_Property_Body) + lv2_atom_pad_size(path_size);
    const uint32_t obj_size = sizeof(LV2_Atom_Object_Body) + prop1 + prop2;
    const uint32_t ev_size = sizeof(LV2_Atom_Event) + obj_size;
    const uint32_t used = sizeof(LV2_Atom) + self->notify_port->atom.size;
    if (used + ev_size > sizeof(LV2_Atom) + self->notify_capacity) return false;

    uint8_t* p = self->notify_buf + used;
    std::memset(p, 0, ev_size);
    LV2_Atom_Event* ev = reinterpret_cast<LV2_Atom_Event*>(p);
    ev->time.frames = 0;
    ev->body.type = self->uris.atom_Object;
    ev->body.size = obj_size;
    LV2_Atom_Object_Body* ob = reinterpret_cast<LV2_Atom_Object_Body*>(ev + 1);
    ob->id = 0;
    ob->otype = self->uris.patch_Set;
    LV2_Atom_Property_Body* p1 = reinterpret_cast<LV2_Atom_Property_Body*>(ob + 1);
    p1->key = self->uris.patch_property;
    p1->context = 0;
    p1->value.type = self->uris.atom_URID;
    p1->value.size = sizeof(uint32_t);
    *reinterpret_cast<uint32_t*>(p1 + 1) = self->uris.json;
    LV2_Atom_Property_Body* p2 = reinterpret_cast<LV2_Atom_Property_Body*>(reinterpret_cast<uint8_t*>(p1) + prop1);
    p2->key = self->uris.patch_value;
    p2->context = 0;
    p2->value.type = self->uris.atom_Path;
    p2->value.size = path_size;
    std::memcpy(p2 + 1, filename, len);
    self->notify_port->atom.size += ev_size;
    return true;
}

bool is_object_type(const PluginURIs* u, uint32_t type)
{
    return type == u->atom_Object || type == u->atom_Blank || type == u->atom_Resource;
}

// ------------------------------------------------------------------ LV2 callbacks

LV2_Handle instantiate(const LV2_Descriptor*, double samplerate, const char*, const LV2_Feature* const* features)
{
    Plugin* self = new Plugin();
    self->samplerate = samplerate;
    for (int i = 0; features && features[i]; ++i) {
        if (!std::strcmp(features[i]->URI, LV2_URID__map)) self->map = static_cast<LV2_URID_Map*>(features[i]->data);
        else if (!std::strcmp(features[i]->URI, LV2_WORKER__schedule)) self->schedule = static_cast<LV2_Worker_Schedule*>(features[i]->data);
        else if (!std::strcmp(features[i]->URI, LV2_LOG__log)) self->log = static_cast<LV2_Log_Log*>(features[i]->data);
    }
    if (!self->map) {
        std::fprintf(stderr, "Error! Missing feature urid:map %s %d\n", __func__, __LINE__);
        delete self;
        return nullptr;
    }
    if (!self->schedule) {
        std::fprintf(stderr, "Error! Missing feature work:schedule %s %d\n", __func__, __LINE__);
        delete self;
        return nullptr;
    }
    map_plugin_uris(self->map, &self->uris);

    const char* hub = std::getenv("AIDAX_HUB");
    self->hub_capacity = hub ? std::atoi(hub) : 0;
    // one-stream mode: this instance's GPU, for its whole life; hub mode: the hub it joins decides (hub_join)
    const int device = place_on_device(self->hub_capacity > 1 ? 0 : 1);
    if (device < 0) {
        std::fprintf(stderr, "Error! %s\n", aidax_last_error());
        delete self;
        return nullptr;
    }
    self->device = device;
    const char* strict = std::getenv("AIDAX_STRICT_REFERENCE_SET");
    self->strict = !(strict && strict[0] == '0');
    // the extension architectures stage whole blocks in LDS: they take pools of <= 256 frames (longer host blocks
    // are chunked in run()); the reference's own model table runs on 8192-frame pools
    self->max_frames = self->strict ? 8192u : 256u;
    // default mode: the instance's own one-stream pool; hub mode: the same call only proves there is a device
    if (aidax_pool_create(1, self->hub_capacity > 1 ? 4 : self->max_frames, samplerate, device, &self->pool) != AIDAX_OK) {
        std::fprintf(stderr, "Error! %s\n", aidax_last_error());
        leave_device(device, self->hub_capacity > 1 ? 0 : 1);
        delete self;
        return nullptr;
    }
    if (self->hub_capacity > 1) {
        aidax_pool_destroy(self->pool);
        self->pool = nullptr;
    }
    self->last_input_size = 0;
    self->loading = true;            // until the host's default-state restore has loaded a model (:318-321)
    self->model = nullptr;
    return self;
}

void connect_port(LV2_Handle instance, uint32_t port, void* data)
{
    Plugin* self = static_cast<Plugin*>(instance);
    switch (port) {
    case IN: self->in = static_cast<const float*>(data); break;
    case OUT_1: self->out_1 = static_cast<float*>(data); break;
    case PLUGIN_CONTROL: self->control_port = static_cast<const LV2_Atom_Sequence*>(data); break;
    case PLUGIN_NOTIFY: self->notify_port = static_cast<LV2_Atom_Sequence*>(data); break;
    case INPUT_SIZE: self->input_size = static_cast<float*>(data); break;
    case LATENCY: self->latency = static_cast<float*>(data); break;
    default:
        if (port < PLUGIN_PORT_COUNT) self->ctl[port] = static_cast<const float*>(data);
        break;
    }
}

void activate(LV2_Handle instance)
{
    Plugin* self = static_cast<Plugin*>(instance);
    if (self->pool) aidax_pool_activate(self->pool, AIDAX_ALL_STREAMS);     // :341-351
    else if (self->hub) aidax_hub_activate(self->hub, self->slot);
}

void deactivate(LV2_Handle) {}

void latch_controls(Plugin* self)
{
    aidax_controls c;
    auto v = [self](int port, float dflt) { return self->ctl[port] ? *self->ctl[port] : dflt; };
    aidax_controls_default(&c);
    c.in_lpf_pc = v(IN_LPF, c.in_lpf_pc);           c.pregain_db = v(PREGAIN, c.pregain_db);
    c.net_bypass = v(NET_BYPASS, c.net_bypass);     c.param1 = v(PARAM1, c.param1);
    c.param2 = v(PARAM2, c.param2);                 c.eq_bypass = v(EQ_BYPASS, c.eq_bypass);
    c.eq_position = v(EQ_POS, c.eq_position);       c.bass_boost_db = v(BASS, c.bass_boost_db);
    c.bass_freq = v(BFREQ, c.bass_freq);            c.mid_boost_db = v(MID, c.mid_boost_db);
    c.mid_freq = v(MFREQ, c.mid_freq);              c.mid_q = v(MIDQ, c.mid_q);
    c.mid_type = v(MTYPE, c.mid_type);              c.treble_boost_db = v(TREBLE, c.treble_boost_db);
    c.treble_freq = v(TFREQ, c.treble_freq);        c.depth_boost_db = v(DEPTH, c.depth_boost_db);
    c.presence_boost_db = v(PRESENCE, c.presence_boost_db);
    c.dc_blocker = v(DCBLOCKER, c.dc_blocker);      c.master_db = v(MASTER, c.master_db);
    c.enabled = v(PLUGIN_ENABLED, c.enabled);
    if (!self->pool) {                                       // hub mode
        self->last_controls = c;
        if (self->hub) {
            aidax_hub_set_controls(self->hub, self->slot, &c);
            if (self->loading != self->last_loading) {
                aidax_hub_set_loading(self->hub, self->slot, self->loading ? 1 : 0);
                self->last_loading = self->loading;
            }
        }
        return;
    }
    if (!self->have_last_controls || std::memcmp(&c, &self->last_controls, sizeof(c)) != 0) {
        aidax_pool_set_controls(self->pool, 0, &c);
        self->last_controls = c;
        self->have_last_controls = true;
    }
    if (self->loading != self->last_loading) {
        aidax_pool_set_loading(self->pool, 0, self->loading ? 1 : 0);
        self->last_loading = self->loading;
    }
}

void run(LV2_Handle instance, uint32_t n_samples)
{
    Plugin* self = static_cast<Plugin*>(instance);
    const PluginURIs* uris = &self->uris;

    if (self->input_size) *self->input_size = static_cast<float>(self->last_input_size);      // :518

    // ---- atom messages (:524-586)
    notify_begin(self);
    if (self->control_port) {
        const LV2_Atom_Sequence* seq = self->control_port;
        const uint8_t* body = reinterpret_cast<const uint8_t*>(&seq->body);
        const uint8_t* end = body + seq->atom.size;
        const uint8_t* p = body + sizeof(LV2_Atom_Sequence_Body);
        while (p + sizeof(LV2_Atom_Event) <= end) {
            const LV2_Atom_Event* ev = reinterpret_cast<const LV2_Atom_Event*>(p);
            p += sizeof(LV2_Atom_Event) + lv2_atom_pad_size(ev->body.size);
            if (!is_object_type(uris, ev->body.type)) {
                plog(self, uris->log_Trace, "Unknown event type %d\n", ev->body.type);
                continue;
            }
            const LV2_Atom_Object* obj = reinterpret_cast<const LV2_Atom_Object*>(&ev->body);
            if (obj->body.otype != uris->patch_Set) {
                plog(self, uris->log_Trace, "Unknown object type %d\n", obj->body.otype);
                continue;
            }
            const LV2_Atom* property = nullptr;
            const LV2_Atom* value = nullptr;
            const uint8_t* ob = reinterpret_cast<const uint8_t*>(&obj->body);
            const uint8_t* oend = ob + obj->atom.size;
            const uint8_t* q = ob + sizeof(LV2_Atom_Object_Body);
            while (q + sizeof(LV2_Atom_Property_Body) <= oend) {
                const LV2_Atom_Property_Body* pr = reinterpret_cast<const LV2_Atom_Property_Body*>(q);
                if (pr->key == uris->patch_property && !property) property = &pr->value;
                else if (pr->key == uris->patch_value && !value) value = &pr->value;
                q += lv2_atom_pad_size(static_cast<uint32_t>(sizeof(LV2_Atom_Property_Body)) + pr->value.size);
            }
            if (!property) { plog(self, uris->log_Trace, "patch:Set message with no property\n"); continue; }
            if (property->type != uris->atom_URID) { plog(self, uris->log_Trace, "patch:Set property is not a URID\n"); continue; }
            if (reinterpret_cast<const LV2_Atom_URID*>(property)->body != uris->json) {
                plog(self, uris->log_Trace, "patch:Set property body is not json\n");
                continue;
            }
            if (!value) { plog(self, uris->log_Trace, "patch:Set message with no value\n"); continue; }
            if (value->type != uris->atom_Path) { plog(self, uris->log_Trace, "patch:Set value is not a Path\n"); continue; }

            plog(self, uris->log_Trace, "Queueing set message\n");
            WorkerLoadMessage msg = { kWorkerLoad, {} };
            std::memcpy(msg.path, value + 1, std::min(value->size, static_cast<uint32_t>(sizeof(msg.path) - 1u)));
            self->schedule->schedule_work(self->schedule->handle, sizeof(msg), &msg);
            self->loading = true;                                                              // :576
        }
    }

    // ---- DSP: control latch, then the whole run() audio section on the GPU (:489-518, :607-659).
    // n_samples == 0 (pre-run) and !enabled (raw copy) are handled inside the pass.
    latch_controls(self);
    int rc = AIDAX_OK;
    if (self->pool) {
        // host blocks longer than the pool's max_frames go through in chunks (state carries over, like any two calls)
        uint32_t done = 0;
        do {
            const uint32_t cnt = std::min(n_samples - done, self->max_frames);
            rc = aidax_pool_process(self->pool, self->in + done, self->out_1 + done, cnt);
            done += cnt;
        } while (rc == AIDAX_OK && done < n_samples);
    } else if (self->hub) {
        // A host block longer than the hub's blocks (offline renders: 4096, 8192 frames) goes through in slices; coming
        // around again closes the period per slice, so each slice returns the result of the slice before it. The hub hands
        // a block's result back only to a block of the SAME length, so every slice of every call must be equally long:
        // the largest divisor of the host's block length that fits (4097 = 17 x 241 -> 241-frame slices; a prime length
        // ends up at one frame per slice — slow, but every sample is delivered, one slice late).
        const uint32_t cap = aidax_hub_max_frames(self->hub);
        if (n_samples <= cap || cap == 0) {
            rc = aidax_hub_run(self->hub, self->slot, self->in, self->out_1, n_samples);
        } else {
            if (self->sliced_n != n_samples) {
                uint32_t k = (n_samples + cap - 1) / cap;
                while (n_samples % k != 0) ++k;                                                // k <= n_samples: ends
                self->sliced_n = n_samples;
                self->slice_len = n_samples / k;
                // (said once per block length, not per call: a length with no divisor between 32 frames and the hub's block costs a
                // pass per few frames — correct, one slice late, and far from real time)
                // run() formats and logs nothing itself (the reference's never does; log:log is not real-time safe): the note travels to
                // the worker thread like a load request does (schedule_work is the one call a run() may make, :576)
                if (self->slice_len < 32u) {
                    const WorkerNoteMessage note = { kWorkerNote, n_samples, k, self->slice_len, cap };
                    self->schedule->schedule_work(self->schedule->handle, sizeof(note), &note);
                }
            }
            const uint32_t slice = self->slice_len;
            for (uint32_t done = 0; done < n_samples && rc == AIDAX_OK; done += slice)
                rc = aidax_hub_run(self->hub, self->slot, self->in + done, self->out_1 + done, slice);
        }
    } else if (n_samples != 0) {
        // hub mode before the first model: the master gain rests at 0 (:306-310), a disabled plugin copies (:612-619)
        if (self->last_controls.enabled > 0.5f) std::memset(self->out_1, 0, sizeof(float) * n_samples);
        else if (self->out_1 != self->in) std::memcpy(self->out_1, self->in, sizeof(float) * n_samples);
    }
    if (rc != AIDAX_OK) {
        // never leave the host's buffer unwritten: silence, or the dry signal when the plugin is disabled
        if (n_samples != 0) {
            if (self->last_controls.enabled > 0.5f) std::memset(self->out_1, 0, sizeof(float) * n_samples);
            else if (self->out_1 != self->in) std::memmove(self->out_1, self->in, sizeof(float) * n_samples);
        }
        if ((self->error_count++ & 1023u) == 0) plog(self, uris->log_Error, "aidax: %s\n", aidax_last_error());
    }
    if (self->latency) *self->latency = self->hub ? static_cast<float>(aidax_hub_latency_frames(self->hub)) : 0.f;
}

void cleanup(LV2_Handle instance)
{
    Plugin* self = static_cast<Plugin*>(instance);
    hub_leave(self->hub, self->slot);
    if (self->hub_capacity <= 1) leave_device(self->device, 1);
    aidax_pool_destroy(self->pool);
    aidax_model_free(self->model);
    delete self;
}

// ---- state (:707-796)
LV2_State_Status restore(LV2_Handle instance, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                         uint32_t, const LV2_Feature* const* features)
{
    Plugin* self = static_cast<Plugin*>(instance);
    size_t size = 0;
    uint32_t type = 0, valflags = 0;
    const void* value = retrieve(handle, self->uris.json, &size, &type, &valflags);
    if (value) {
        plog(self, self->uris.log_Note, "Restoring file %s\n", static_cast<const char*>(value));
        WorkerLoadMessage msg = { kWorkerLoad, {} };
        LV2_State_Map_Path* map_path = nullptr;
        LV2_State_Free_Path* free_path = nullptr;
        for (int i = 0; features && features[i]; ++i) {
            if (!std::strcmp(features[i]->URI, LV2_STATE__mapPath)) map_path = static_cast<LV2_State_Map_Path*>(features[i]->data);
            else if (!std::strcmp(features[i]->URI, LV2_STATE__freePath)) free_path = static_cast<LV2_State_Free_Path*>(features[i]->data);
        }
        if (map_path) {
            char* apath = map_path->absolute_path(map_path->handle, static_cast<const char*>(value));
            std::memcpy(msg.path, apath, std::min(std::strlen(apath), sizeof(msg.path) - 1u));
            if (free_path) free_path->free_path(free_path->handle, apath);
            else std::free(apath);
        } else {
            std::memcpy(msg.path, value, std::min(size, sizeof(msg.path) - 1u));
        }
        self->schedule->schedule_work(self->schedule->handle, sizeof(msg), &msg);
    }
    return LV2_STATE_SUCCESS;
}

LV2_State_Status save(LV2_Handle instance, LV2_State_Store_Function store, LV2_State_Handle handle,
                      uint32_t, const LV2_Feature* const* features)
{
    Plugin* self = static_cast<Plugin*>(instance);
    if (!self->model) return LV2_STATE_SUCCESS;              // nothing loaded yet (:772-774)
    LV2_State_Map_Path* map_path = nullptr;
    for (int i = 0; features && features[i]; ++i)
        if (!std::strcmp(features[i]->URI, LV2_STATE__mapPath)) map_path = static_cast<LV2_State_Map_Path*>(features[i]->data);
    if (!map_path) return LV2_STATE_ERR_NO_FEATURE;          // :793-795
    char* apath = map_path->abstract_path(map_path->handle, aidax_model_path(self->model));
    store(handle, self->uris.json, apath, std::strlen(apath) + 1, self->uris.atom_Path,
          LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
    std::free(apath);
    return LV2_STATE_SUCCESS;
}

// ---- worker (:807-893)
LV2_Worker_Status work(LV2_Handle instance, LV2_Worker_Respond_Function respond, LV2_Worker_Respond_Handle handle,
                       uint32_t, const void* data)
{
    Plugin* self = static_cast<Plugin*>(instance);
    const WorkerMessage* msg = static_cast<const WorkerMessage*>(data);
    switch (msg->type) {
    case kWorkerLoad: {
        const char* path = static_cast<const WorkerLoadMessage*>(data)->path;
        aidax_model* m = nullptr;
        if (aidax_model_load(path, &m) != AIDAX_OK) {
            // no reply: the old model keeps playing but `loading` stays set, so the master ramps to 0 (:576, :654)
            plog(self, self->uris.log_Error, "%s\n", aidax_last_error());
            return LV2_WORKER_SUCCESS;
        }
        aidax_model_info_t info;
        aidax_model_info(m, &info);
        if (self->strict && !info.in_reference_set) {            // custom_model_creator fails (:1025-1026)
            plog(self, self->uris.log_Error, "Error loading model: Unable to identify a known model architecture!\n");
            aidax_model_free(m);
            return LV2_WORKER_SUCCESS;
        }
        // the device half of loadModelFromPath (:1034-1079), still on the worker: upload, reset, PARAM smoothers
        // around the playing model's targets (:822-825), 2048-zero warm-up
        WorkerApplyMessage reply = { kWorkerApply, m, nullptr, nullptr, -1 };
        bool ready;
        if (self->pool) ready = aidax_pool_prepare_model(self->pool, m, AIDAX_START_WARMUP, &reply.staged) == AIDAX_OK;
        else ready = hub_join(self, m, &reply.hub, &reply.slot);
        if (!ready) {
            plog(self, self->uris.log_Error, "aidax: %s\n", aidax_last_error());
            aidax_model_free(m);
            return LV2_WORKER_SUCCESS;
        }
        self->last_input_size = info.input_size;                 // cached for the ModelInSize port (:1082)
        plog(self, self->uris.log_Note, "Successfully loaded json file: %s\n", path);
        respond(handle, sizeof(reply), &reply);
        return LV2_WORKER_SUCCESS;
    }
    case kWorkerNote: {
        const WorkerNoteMessage* nt = static_cast<const WorkerNoteMessage*>(data);
        plog(self, self->uris.log_Note, "aidax: hub mode: a host block of %u frames goes through in %u slices of %u (no larger divisor fits the hub's %u-frame blocks): "
             "use a block length with a divisor in [32, %u] or raise AIDAX_HUB_FRAMES\n", nt->n_samples, nt->slices, nt->slice_len, nt->cap, nt->cap);
        return LV2_WORKER_SUCCESS;
    }
    case kWorkerFree: {
        const WorkerApplyMessage* old = static_cast<const WorkerApplyMessage*>(data);
        aidax_model_free(old->model);                             // freeModel (:838-840)
        aidax_staged_free(old->staged);                           // the device buffers the swap retired
        hub_leave(old->hub, old->slot);
        return LV2_WORKER_SUCCESS;
    }
    case kWorkerApply:
        break;                                                    // should not happen
    }
    return LV2_WORKER_ERR_UNKNOWN;
}

LV2_Worker_Status work_response(LV2_Handle instance, uint32_t, const void* data)
{
    Plugin* self = static_cast<Plugin*>(instance);
    const WorkerMessage* msg = static_cast<const WorkerMessage*>(data);
    if (msg->type != kWorkerApply) return LV2_WORKER_ERR_UNKNOWN;
    const WorkerApplyMessage* apply = static_cast<const WorkerApplyMessage*>(data);

    // swap (:868-875): handles only. The old model, the buffers the swap retires and (hub mode) the old seat go
    // back to the worker for deletion.
    WorkerApplyMessage reply = { kWorkerFree, self->model, apply->staged, self->hub, self->slot };
    if (self->pool) {
        if (aidax_pool_commit_model(self->pool, apply->staged) != AIDAX_OK) {
            // cannot happen with a staged object of this pool; keep the old model, give the new one back
            plog(self, self->uris.log_Error, "aidax: %s\n", aidax_last_error());
            reply.model = apply->model;
            self->schedule->schedule_work(self->schedule->handle, sizeof(reply), &reply);
            return LV2_WORKER_SUCCESS;
        }
    } else {
        // the plugin's own members (biquads, gain smoothers: rt-neural-generic.h:311-317) stay what they are across a
        // swap: the new seat adopts them from the old one, on the device, behind the old seat's last block
        if (self->hub && aidax_hub_adopt(apply->hub, apply->slot, self->hub, self->slot) != AIDAX_OK)
            plog(self, self->uris.log_Error, "aidax: %s\n", aidax_last_error());
        self->hub = apply->hub;
        self->slot = apply->slot;
        self->last_loading = false;                           // a freshly attached stream is not loading
    }
    self->model = apply->model;
    self->schedule->schedule_work(self->schedule->handle, sizeof(reply), &reply);
    plog(self, self->uris.log_Trace, "New model in use\n");
    notify_set_file(self, aidax_model_path(self->model));    // report change to host/ui (:880-887)
    self->loading = false;                                    // :889
    self->last_loading = false;                               // the commit cleared the pool's flag as well
    plog(self, self->uris.log_Trace, "loading = false\n");
    return LV2_WORKER_SUCCESS;
}

const void* extension_data(const char* uri)
{
    static const LV2_State_Interface state = { save, restore };
    if (!std::strcmp(uri, LV2_STATE__interface)) return &state;
    static const LV2_Worker_Interface worker = { work, work_response, nullptr };
    if (!std::strcmp(uri, LV2_WORKER__interface)) return &worker;
    return nullptr;
}

const LV2_Descriptor kDescriptor = {
    PLUGIN_URI, instantiate, connect_port, activate, run, deactivate, cleanup, extension_data
};

}  // namespace

extern "C" LV2_SYMBOL_EXPORT const LV2_Descriptor* lv2_descriptor(uint32_t index)
{
    return index == 0 ? &kDescriptor : nullptr;
}
