"""A minimal mock LV2 host in ctypes for the plugin shell tests: feature array
(urid:map, work:schedule, state:mapPath/freePath), a worker queue the test pumps by
hand, atom sequence buffers for the CONTROL / NOTIFY ports, and 21 control ports.
Struct layouts follow the public LV2 C ABI (see aidadsp-lv2_amd/lv2/lv2_min.h)."""
import ctypes as C
import os
import struct

import numpy as np

if not os.environ.get("AIDAX_NO_TORCH"):   # one HIP runtime per test process: see aidadsp-lv2_amd/binding.py lib()
    try:
        import torch  # noqa: F401
    except Exception:
        pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "aidadsp-lv2_amd", "lv2", "rt-neural-generic.so")
PLUGIN_URI = b"http://aidadsp.cc/plugins/aidadsp-bundle/rt-neural-generic"

PORTS = ["IN", "OUT", "CONTROL", "NOTIFY", "ANTIALIASING", "PREGAIN", "NETBYPASS", "PARAM1", "PARAM2",
         "EQBYPASS", "EQPOS", "BASS", "BFREQ", "MID", "MFREQ", "MIDQ", "MTYPE", "TREBLE", "TFREQ",
         "DEPTH", "PRESENCE", "DCBLOCKER", "MASTER", "ModelInSize", "enabled"]        # rt-neural-generic.ttl:61-313
DEFAULTS = {"ANTIALIASING": 66.216, "PREGAIN": 0, "NETBYPASS": 0, "PARAM1": 0, "PARAM2": 0, "EQBYPASS": 0,
            "EQPOS": 0, "BASS": 0, "BFREQ": 305.0, "MID": 0, "MFREQ": 750.0, "MIDQ": 0.707, "MTYPE": 0,
            "TREBLE": 0, "TFREQ": 2000.0, "DEPTH": 0, "PRESENCE": 0, "DCBLOCKER": 1, "MASTER": 0,
            "ModelInSize": 0, "enabled": 1}
# control-port symbol -> field of the oracle / C-ABI controls struct
FIELD = {"ANTIALIASING": "in_lpf_pc", "PREGAIN": "pregain_db", "NETBYPASS": "net_bypass", "PARAM1": "param1",
         "PARAM2": "param2", "EQBYPASS": "eq_bypass", "EQPOS": "eq_position", "BASS": "bass_boost_db",
         "BFREQ": "bass_freq", "MID": "mid_boost_db", "MFREQ": "mid_freq", "MIDQ": "mid_q", "MTYPE": "mid_type",
         "TREBLE": "treble_boost_db", "TFREQ": "treble_freq", "DEPTH": "depth_boost_db",
         "PRESENCE": "presence_boost_db", "DCBLOCKER": "dc_blocker", "MASTER": "master_db", "enabled": "enabled"}


class Feature(C.Structure):
    _fields_ = [("URI", C.c_char_p), ("data", C.c_void_p)]


INSTANTIATE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_double, C.c_char_p, C.POINTER(C.POINTER(Feature)))
CONNECT = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p)
VOIDH = C.CFUNCTYPE(None, C.c_void_p)
RUN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)
EXTDATA = C.CFUNCTYPE(C.c_void_p, C.c_char_p)


class Descriptor(C.Structure):
    _fields_ = [("URI", C.c_char_p), ("instantiate", INSTANTIATE), ("connect_port", CONNECT), ("activate", VOIDH),
                ("run", RUN), ("deactivate", VOIDH), ("cleanup", VOIDH), ("extension_data", EXTDATA)]


MAPFN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_char_p)


class UridMap(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("map", MAPFN)]


SCHEDFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)


class Schedule(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("schedule_work", SCHEDFN)]


RESPOND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)
WORK = C.CFUNCTYPE(C.c_int, C.c_void_p, RESPOND, C.c_void_p, C.c_uint32, C.c_void_p)
WORKRESP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)


class WorkerIface(C.Structure):
    _fields_ = [("work", WORK), ("work_response", WORKRESP), ("end_run", C.c_void_p)]


STORE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32)
RETRIEVE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
SAVE = C.CFUNCTYPE(C.c_int, C.c_void_p, STORE, C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(Feature)))
RESTORE = C.CFUNCTYPE(C.c_int, C.c_void_p, RETRIEVE, C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(Feature)))


class StateIface(C.Structure):
    _fields_ = [("save", SAVE), ("restore", RESTORE)]


PATHFN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p)      # returns malloc'd char*


class MapPath(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("abstract_path", PATHFN), ("absolute_path", PATHFN)]


FREEFN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class FreePath(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("free_path", FREEFN)]


_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def _cstr_malloc(b: bytes) -> int:
    p = _libc.malloc(len(b) + 1)
    C.memmove(p, b + b"\0", len(b) + 1)
    return p


def features_array(feats):
    arr = (C.POINTER(Feature) * (len(feats) + 1))()
    for i, f in enumerate(feats):
        arr[i] = C.pointer(f)
    arr[len(feats)] = None
    return arr


class Host:
    """One plugin instance plus everything the host side must own."""

    def __init__(self, samplerate=48000.0, bundle_dir=None, with_map=True, with_schedule=True, block=256, so=None):
        self.lib = C.CDLL(so or SO)
        self.lib.lv2_descriptor.restype = C.POINTER(Descriptor)
        self.lib.lv2_descriptor.argtypes = [C.c_uint32]
        self.desc = self.lib.lv2_descriptor(0).contents
        self.bundle_dir = bundle_dir or ROOT
        self.urids = {}
        self.rev = {}
        self.work_queue = []          # (bytes) messages scheduled by the plugin
        self.responses = []           # (bytes) responses from work()
        self.freed = []

        def _map(_h, uri):
            u = uri.decode()
            if u not in self.urids:
                self.urids[u] = len(self.urids) + 1
                self.rev[self.urids[u]] = u
            return self.urids[u]

        def _sched(_h, size, data):
            self.work_queue.append(C.string_at(data, size))
            return 0

        self._map_cb = MAPFN(_map)
        self._sched_cb = SCHEDFN(_sched)
        self.urid_map = UridMap(None, self._map_cb)
        self.schedule = Schedule(None, self._sched_cb)

        def _abs(_h, p):
            return _cstr_malloc(os.path.join(self.bundle_dir, p.decode()).encode())

        def _abst(_h, p):
            return _cstr_malloc(os.path.relpath(p.decode(), self.bundle_dir).encode())

        self.free_calls = 0

        def _free(_h, p):
            self.free_calls += 1
            _libc.free(p)

        self._abs_cb, self._abst_cb, self._free_cb = PATHFN(_abs), PATHFN(_abst), FREEFN(_free)
        self.map_path = MapPath(None, self._abst_cb, self._abs_cb)
        self.free_path = FreePath(None, self._free_cb)

        feats = []
        if with_map:
            feats.append(Feature(b"http://lv2plug.in/ns/ext/urid#map", C.cast(C.pointer(self.urid_map), C.c_void_p)))
        if with_schedule:
            feats.append(Feature(b"http://lv2plug.in/ns/ext/worker#schedule", C.cast(C.pointer(self.schedule), C.c_void_p)))
        self._feats = feats
        self._feat_arr = features_array(feats)
        self.handle = self.desc.instantiate(C.byref(self.desc), samplerate, self.bundle_dir.encode(), self._feat_arr)
        if not self.handle:
            return
        # ports
        self.block = block
        self.audio_in = np.zeros(8192, np.float32)
        self.audio_out = np.zeros(8192, np.float32)
        self.ctl = {name: C.c_float(DEFAULTS[name]) for name in PORTS[4:]}
        self.control_buf = (C.c_uint8 * 4096)()
        self.notify_buf = (C.c_uint8 * 4096)()
        self.desc.connect_port(self.handle, 0, self.audio_in.ctypes.data_as(C.c_void_p))
        self.desc.connect_port(self.handle, 1, self.audio_out.ctypes.data_as(C.c_void_p))
        self.desc.connect_port(self.handle, 2, C.cast(self.control_buf, C.c_void_p))
        self.desc.connect_port(self.handle, 3, C.cast(self.notify_buf, C.c_void_p))
        for i, name in enumerate(PORTS[4:], start=4):
            self.desc.connect_port(self.handle, i, C.cast(C.pointer(self.ctl[name]), C.c_void_p))
        self.worker = C.cast(self.desc.extension_data(b"http://lv2plug.in/ns/ext/worker#interface"), C.POINTER(WorkerIface)).contents
        self.state = C.cast(self.desc.extension_data(b"http://lv2plug.in/ns/ext/state#interface"), C.POINTER(StateIface)).contents
        self.clear_control()

    # ---- atoms
    def urid(self, uri):
        return self._map_cb(None, uri.encode())

    def clear_control(self):
        seq = struct.pack("<IIII", 8, self.urid("http://lv2plug.in/ns/ext/atom#Sequence"), 0, 0)
        C.memmove(self.control_buf, seq, len(seq))

    def send_patch_set(self, path, prop_uri=PLUGIN_URI.decode() + "#json", value_type="http://lv2plug.in/ns/ext/atom#Path",
                       otype="http://lv2plug.in/ns/ext/patch#Set"):
        """One patch:Set{property, value} event at frame 0 in the CONTROL sequence."""
        pad = lambda n: (n + 7) & ~7
        pb = path.encode() + b"\0"
        prop1 = struct.pack("<IIIII", self.urid("http://lv2plug.in/ns/ext/patch#property"), 0, 4,
                            self.urid("http://lv2plug.in/ns/ext/atom#URID"), self.urid(prop_uri)) + b"\0" * 4
        prop2 = struct.pack("<IIII", self.urid("http://lv2plug.in/ns/ext/patch#value"), 0, len(pb), self.urid(value_type)) + pb
        prop2 += b"\0" * (pad(len(prop2)) - len(prop2))
        obj_body = struct.pack("<II", 0, self.urid(otype)) + prop1 + prop2
        ev = struct.pack("<qII", 0, len(obj_body), self.urid("http://lv2plug.in/ns/ext/atom#Object")) + obj_body
        ev += b"\0" * (pad(len(ev)) - len(ev))
        seq = struct.pack("<IIII", 8 + len(ev), self.urid("http://lv2plug.in/ns/ext/atom#Sequence"), 0, 0) + ev
        C.memmove(self.control_buf, seq, len(seq))

    def read_notify(self):
        """Decode the NOTIFY sequence -> list of (otype_uri, {key_uri: (type_uri, bytes)})."""
        raw = bytes(self.notify_buf)
        size, typ = struct.unpack_from("<II", raw, 0)
        out = []
        pos, end = 16, 8 + size
        while pos + 16 <= end:
            _frames, bsize, btype = struct.unpack_from("<qII", raw, pos)
            body = raw[pos + 16: pos + 16 + bsize]
            _id, otype = struct.unpack_from("<II", body, 0)
            props, q = {}, 8
            while q + 16 <= len(body):
                key, _ctx, vsize, vtype = struct.unpack_from("<IIII", body, q)
                props[self.rev.get(key, key)] = (self.rev.get(vtype, vtype), body[q + 16: q + 16 + vsize])
                q += (16 + vsize + 7) & ~7
            out.append((self.rev.get(otype, otype), props))
            pos += 16 + ((bsize + 7) & ~7)
        return out

    # ---- run loop pieces
    def run(self, x):
        n = len(x)
        self.audio_in[:n] = x
        # host convention for output atom ports: atom.size = capacity before run()
        struct.pack_into("<II", self.notify_buf, 0, len(self.notify_buf) - 8, 0)
        self.desc.run(self.handle, n)
        self.clear_control()
        return self.audio_out[:n].copy()

    def pump_worker(self):
        """Run every scheduled job on the 'worker thread' (here: inline), collecting responses."""
        def _respond(_h, size, data):
            self.responses.append(C.string_at(data, size))
            return 0
        cb = RESPOND(_respond)
        n = 0
        while self.work_queue:
            msg = self.work_queue.pop(0)
            buf = C.create_string_buffer(msg, len(msg))
            rc = self.worker.work(self.handle, cb, None, len(msg), C.cast(buf, C.c_void_p))
            assert rc == 0, rc
            n += 1
        return n

    def deliver_responses(self):
        """What a host does right after run(): hand work() replies to work_response()."""
        n = 0
        while self.responses:
            r = self.responses.pop(0)
            buf = C.create_string_buffer(r, len(r))
            rc = self.worker.work_response(self.handle, len(r), C.cast(buf, C.c_void_p))
            assert rc == 0, rc
            n += 1
        return n

    def restore(self, abstract_path, with_map_path=True, with_free_path=True):
        val = C.create_string_buffer(abstract_path.encode() + b"\0")
        key_json = self.urid(PLUGIN_URI.decode() + "#json")

        def _retrieve(_h, key, size, typ, flags):
            if key != key_json:
                return None
            size[0] = len(val.raw)
            typ[0] = self.urid("http://lv2plug.in/ns/ext/atom#Path")
            flags[0] = 3
            return C.cast(val, C.c_void_p).value
        cb = RETRIEVE(_retrieve)
        feats = []
        if with_map_path:
            feats.append(Feature(b"http://lv2plug.in/ns/ext/state#mapPath", C.cast(C.pointer(self.map_path), C.c_void_p)))
        if with_free_path:
            feats.append(Feature(b"http://lv2plug.in/ns/ext/state#freePath", C.cast(C.pointer(self.free_path), C.c_void_p)))
        arr = features_array(feats)
        return self.state.restore(self.handle, cb, None, 0, arr)

    def save(self, with_map_path=True):
        stored = []

        def _store(_h, key, value, size, typ, flags):
            stored.append((self.rev.get(key, key), C.string_at(value, size), self.rev.get(typ, typ), flags))
            return 0
        cb = STORE(_store)
        feats = [Feature(b"http://lv2plug.in/ns/ext/state#mapPath", C.cast(C.pointer(self.map_path), C.c_void_p))] if with_map_path else []
        arr = features_array(feats)
        rc = self.state.save(self.handle, cb, None, 0, arr)
        return rc, stored

    def controls(self, **kw):
        for k, v in kw.items():
            self.ctl[k].value = v

    def close(self):
        if self.handle:
            self.desc.cleanup(self.handle)
            self.handle = None
