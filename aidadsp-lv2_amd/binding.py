"""ctypes binding of include/aidax.h (no logic beyond argument marshalling)."""
from __future__ import annotations

import ctypes as C
import os
import re
import weakref
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_HEADER = os.path.join(_ROOT, "include", "aidax.h")

ALL_STREAMS = -1
START_WARMUP, START_RESET = 0, 1
_fp = C.POINTER(C.c_float)


class AidaxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"aidax error {code}: {msg}")
        self.code = code


class ModelInfo(C.Structure):
    _fields_ = [("cell", C.c_int32), ("hidden", C.c_int32), ("input_size", C.c_int32),
                ("n_rnn_layers", C.c_int32), ("input_skip", C.c_int32),
                ("input_gain", C.c_float), ("output_gain", C.c_float), ("samplerate", C.c_float),
                ("n_golden", C.c_int32), ("in_reference_set", C.c_int32), ("n_weights", C.c_uint64)]


CONTROL_FIELDS = ("in_lpf_pc", "pregain_db", "net_bypass", "param1", "param2", "eq_bypass",
                  "eq_position", "bass_boost_db", "bass_freq", "mid_boost_db", "mid_freq", "mid_q",
                  "mid_type", "treble_boost_db", "treble_freq", "depth_boost_db",
                  "presence_boost_db", "dc_blocker", "master_db", "enabled")


class StreamDsp(C.Structure):
    _fields_ = [("z", (C.c_double * 2) * 7), ("pre_mem", C.c_float), ("master_mem", C.c_float), ("pre_target", C.c_float),
                ("master_target", C.c_float), ("param_target", C.c_float * 2)]


class Controls(C.Structure):
    _fields_ = [(n, C.c_float) for n in CONTROL_FIELDS]


def lib_path() -> str:
    # AIDAX_LIB: another build of the library (A/B measurements of two builds in one gpurun call)
    return os.environ.get("AIDAX_LIB") or os.path.join(_HERE, "lib", "libaidax_hip.so")


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library. Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # Python processes that also use torch (bench.py, some tests) must end up with ONE HIP/HSA
    # runtime: torch ships its own copy, and a second runtime initialised later finds no GPU.
    # Importing torch first makes the loader resolve libamdhip64.so.7 to the copy torch loaded.
    # (AIDAX_NO_TORCH=1: processes that never touch torch, e.g. tests/rt_audit.py, skip that.)
    if not os.environ.get("AIDAX_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `make` (or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
    L.aidax_last_error.restype = C.c_char_p
    L.aidax_version.restype = C.c_char_p
    L.aidax_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.aidax_model_load_memory.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(vp)]
    L.aidax_model_info.argtypes = [vp, C.POINTER(ModelInfo)]
    L.aidax_model_path.argtypes = [vp]
    L.aidax_model_path.restype = C.c_char_p
    L.aidax_model_golden.argtypes = [vp, _fp, _fp, u32]
    L.aidax_model_free.argtypes = [vp]
    L.aidax_model_free.restype = None
    L.aidax_controls_default.argtypes = [C.POINTER(Controls)]
    L.aidax_controls_default.restype = None
    L.aidax_biquad_design.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.aidax_db_to_coeff.argtypes = [C.c_float]
    L.aidax_db_to_coeff.restype = C.c_float
    L.aidax_lpf_fc.argtypes = [C.c_float]
    L.aidax_lpf_fc.restype = C.c_float
    L.aidax_device_count.argtypes = [C.POINTER(C.c_int)]
    L.aidax_pick_device.argtypes = [C.c_char_p, C.c_int, C.POINTER(u32), C.POINTER(C.c_int)]
    L.aidax_pick_hub.argtypes = [C.POINTER(C.c_int), C.POINTER(u32), C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(u32),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.aidax_many_streams_form.argtypes = [C.c_int, C.c_int, u32, C.c_int]
    L.aidax_many_streams_form.restype = C.c_int
    L.aidax_many_streams_form_at.argtypes = [C.c_int, C.c_int, u32, C.c_int, u32]
    L.aidax_many_streams_form_at.restype = C.c_int
    L.aidax_model_conv_form.argtypes = [C.c_void_p]
    L.aidax_model_conv_form.restype = C.c_int
    L.aidax_pool_create.argtypes = [u32, u32, C.c_double, C.c_int, C.POINTER(vp)]
    L.aidax_pool_destroy.argtypes = [vp]
    L.aidax_pool_destroy.restype = None
    L.aidax_pool_streams.argtypes = [vp]
    L.aidax_pool_streams.restype = u32
    L.aidax_pool_set_model.argtypes = [vp, vp, C.c_int]
    L.aidax_pool_prepare_model.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.aidax_pool_commit_model.argtypes = [vp, vp]
    L.aidax_staged_free.argtypes = [vp]
    L.aidax_staged_free.restype = None
    L.aidax_pool_set_loading.argtypes = [vp, i32, C.c_int]
    L.aidax_pool_set_controls.argtypes = [vp, i32, C.POINTER(Controls)]
    L.aidax_pool_activate.argtypes = [vp, i32]
    L.aidax_pool_process.argtypes = [vp, _fp, _fp, u32]
    L.aidax_pool_submit.argtypes = [vp, _fp, u32]
    L.aidax_pool_collect.argtypes = [vp, _fp, u32]
    L.aidax_pool_submit_to.argtypes = [vp, _fp, _fp, u32]
    L.aidax_pool_register_host.argtypes = [vp, C.c_void_p, C.c_size_t]
    L.aidax_pool_unregister_host.argtypes = [vp, C.c_void_p]
    L.aidax_pool_process_device.argtypes = [vp, vp, vp, u32, vp]
    L.aidax_pool_sync.argtypes = [vp]
    L.aidax_model_self_test.argtypes = [vp, C.c_int, C.POINTER(i32), _fp, _fp]
    L.aidax_model_forward.argtypes = [vp, C.c_int, _fp, _fp, u32, C.c_int]
    L.aidax_pool_read_state.argtypes = [vp, u32, C.c_int, _fp, _fp, u32]
    L.aidax_pool_kernel_name.argtypes = [vp]
    L.aidax_pool_kernel_name.restype = C.c_char_p
    L.aidax_pool_reset_stream.argtypes = [vp, u32, C.c_int]
    L.aidax_hub_create.argtypes = [u32, u32, C.c_double, C.c_int, C.POINTER(vp)]
    L.aidax_hub_destroy.argtypes = [vp]
    L.aidax_hub_destroy.restype = None
    L.aidax_hub_set_model.argtypes = [vp, vp, C.c_int]
    L.aidax_hub_attach.argtypes = [vp, C.POINTER(i32)]
    L.aidax_hub_detach.argtypes = [vp, i32]
    L.aidax_hub_set_controls.argtypes = [vp, i32, C.POINTER(Controls)]
    L.aidax_hub_run.argtypes = [vp, i32, _fp, _fp, u32]
    L.aidax_hub_set_loading.argtypes = [vp, i32, C.c_int]
    L.aidax_hub_activate.argtypes = [vp, i32]
    L.aidax_hub_latency_frames.argtypes = [vp]
    L.aidax_hub_latency_frames.restype = u32
    L.aidax_hub_attached.argtypes = [vp]
    L.aidax_hub_attached.restype = u32
    L.aidax_hub_max_frames.argtypes = [vp]
    L.aidax_hub_max_frames.restype = u32
    L.aidax_hub_attach_successor.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.aidax_hub_adopt.argtypes = [vp, C.c_int32, vp, C.c_int32]
    L.aidax_pool_export_stream_dsp.argtypes = [vp, u32, C.POINTER(StreamDsp)]
    L.aidax_pool_import_stream_dsp.argtypes = [vp, u32, C.POINTER(StreamDsp)]
    L.aidax_hub_launches.argtypes = [vp]
    L.aidax_hub_launches.restype = C.c_uint64
    L.aidax_hub_deadline_launches.argtypes = [vp]
    L.aidax_hub_deadline_launches.restype = C.c_uint64
    L.aidax_hub_faults_unmapped.argtypes = [vp]
    L.aidax_hub_faults_unmapped.restype = C.c_uint64
    L.aidax_hub_set_deadline_us.argtypes = [vp, C.c_int64]
    L.aidax_hub_flush.argtypes = [vp]
    _lib = L
    return L


def declared_symbols() -> list:
    """Every AIDAX_API function name declared in include/aidax.h."""
    with open(_HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"AIDAX_API[^;(]*?\b(aidax_[a-z_0-9]+)\s*\(", text)))


def _check(rc: int):
    if rc < 0:
        raise AidaxError(rc, lib().aidax_last_error().decode(errors="replace"))
    return rc


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def default_controls(**kw) -> Controls:
    c = Controls()
    lib().aidax_controls_default(C.byref(c))
    for k, v in kw.items():
        if k not in CONTROL_FIELDS:
            raise KeyError(k)
        setattr(c, k, v)
    return c


def biquad_design(kind: int, fc: float, q: float, gain_db: float) -> np.ndarray:
    out = (C.c_double * 5)()
    _check(lib().aidax_biquad_design(kind, fc, q, gain_db, out))
    return np.array(list(out), np.float64)


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().aidax_device_count(C.byref(n)))
    return n.value


def pick_device(spec: Optional[str], count: int, load=None) -> int:
    """The placement rule (pure): least-loaded device among those `spec` names ("auto", "0-3,6", None = device 0)."""
    arr = None
    if load is not None:
        arr = (C.c_uint32 * len(load))(*load)
    out = C.c_int(-1)
    _check(lib().aidax_pick_device(spec.encode() if spec is not None else None, count, arr, C.byref(out)))
    return out.value


def pick_hub(hub_devices, hub_free_seats, current_device: int, spec: Optional[str], count: int, load=None):
    """-> (index of the hub to join or -1 = open a new one, device)"""
    n = len(hub_devices)
    dev = (C.c_int * max(n, 1))(*hub_devices)
    free = (C.c_uint32 * max(n, 1))(*hub_free_seats)
    arr = (C.c_uint32 * count)(*load) if load is not None else None
    idx, out = C.c_int(-2), C.c_int(-1)
    _check(lib().aidax_pick_hub(dev, free, n, current_device, spec.encode() if spec is not None else None, count, arr,
                                C.byref(idx), C.byref(out)))
    return idx.value, out.value


def many_streams_form(cell: int, hidden: int, n_streams: int, compute_units: int = 256, max_frames: int = 256) -> int:
    """aidax_many_streams_form_at: 0 per-stream forms, 1 k_quad, 2 the 16-stream matrix-core kernels"""
    return int(lib().aidax_many_streams_form_at(cell, hidden, n_streams, compute_units, max_frames))


def db_to_coeff(db: float) -> float:
    return float(lib().aidax_db_to_coeff(C.c_float(db)))


def lpf_fc(pc: float) -> float:
    return float(lib().aidax_lpf_fc(C.c_float(pc)))


class Model:
    """aidax_model: the json-side half of the reference's DynamicModel."""

    def __init__(self, path: Optional[str] = None, text: Optional[str] = None, label: str = "<memory>"):
        h = C.c_void_p()
        if path is not None:
            _check(lib().aidax_model_load(path.encode(), C.byref(h)))
        else:
            b = text.encode()
            _check(lib().aidax_model_load_memory(b, len(b), label.encode(), C.byref(h)))
        self.h = h

    @property
    def info(self) -> ModelInfo:
        i = ModelInfo()
        _check(lib().aidax_model_info(self.h, C.byref(i)))
        return i

    @property
    def path(self) -> str:
        return lib().aidax_model_path(self.h).decode()

    @property
    def conv_form(self) -> int:
        """aidax_model_conv_form: 0 not a conv stack, 1 k_conv, 2 k_conv_mfma, 3 k_conv_ms, 4 k_conv_ms + k_conv_st for full blocks"""
        return int(lib().aidax_model_conv_form(self.h))

    def golden(self):
        n = self.info.n_golden
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        lib().aidax_model_golden(self.h, a.ctypes.data_as(_fp), b.ctypes.data_as(_fp), n)
        return a, b

    def self_test(self, device: int = 0):
        n_err, max_err = C.c_int32(0), C.c_float(0)
        out = np.zeros(self.info.n_golden, np.float32)
        _check(lib().aidax_model_self_test(self.h, device, C.byref(n_err), C.byref(max_err), out.ctypes.data_as(_fp)))
        return int(n_err.value), float(max_err.value), out

    def forward(self, X, device: int = 0, unit_gains: bool = False) -> np.ndarray:
        X = _f32(X).reshape(-1, self.info.input_size)
        y = np.zeros(X.shape[0], np.float32)
        _check(lib().aidax_model_forward(self.h, device, X.ctypes.data_as(_fp), y.ctypes.data_as(_fp),
                                         X.shape[0], 1 if unit_gains else 0))
        return y

    def close(self):
        if self.h:
            lib().aidax_model_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# the pools and hubs that are open (test harnesses close what a failed test left behind: a pool that holds a device's right to
# the chained kernels would otherwise change the kernel every later pool of the process gets)
live_handles = weakref.WeakSet()


class Pool:
    """aidax_pool: N plugin instances' DSP state on one GPU."""

    def __init__(self, n_streams: int, max_frames: int = 256, samplerate: float = 48000.0, device: int = 0):
        h = C.c_void_p()
        _check(lib().aidax_pool_create(n_streams, max_frames, samplerate, device, C.byref(h)))
        self.h = h
        live_handles.add(self)
        self.n_streams = n_streams
        self.max_frames = max_frames

    def set_model(self, m: Optional[Model], start_mode: int = START_WARMUP):
        _check(lib().aidax_pool_set_model(self.h, m.h if m is not None else None, start_mode))

    def prepare_model(self, m: Optional[Model], start_mode: int = START_WARMUP) -> C.c_void_p:
        """worker half of a model swap: returns the staged handle for commit_model / staged_free"""
        sg = C.c_void_p()
        _check(lib().aidax_pool_prepare_model(self.h, m.h if m is not None else None, start_mode, C.byref(sg)))
        return sg

    def commit_model(self, staged: C.c_void_p):
        """audio half: swaps the staged model in; `staged` then holds what was retired (free it with staged_free)"""
        _check(lib().aidax_pool_commit_model(self.h, staged))

    @staticmethod
    def staged_free(staged: C.c_void_p):
        lib().aidax_staged_free(staged)

    def set_controls(self, c: Controls, stream: int = ALL_STREAMS):
        _check(lib().aidax_pool_set_controls(self.h, stream, C.byref(c)))

    def set_loading(self, loading: bool, stream: int = ALL_STREAMS):
        _check(lib().aidax_pool_set_loading(self.h, stream, 1 if loading else 0))

    def activate(self, stream: int = ALL_STREAMS):
        _check(lib().aidax_pool_activate(self.h, stream))

    def process(self, x: np.ndarray) -> np.ndarray:
        x = _f32(x)
        assert x.ndim == 2 and x.shape[0] == self.n_streams
        out = np.empty_like(x)
        _check(lib().aidax_pool_process(self.h, x.ctypes.data_as(_fp), out.ctypes.data_as(_fp), x.shape[1]))
        return out

    def submit(self, x: np.ndarray):
        x = _f32(x)
        assert x.ndim == 2 and x.shape[0] == self.n_streams
        _check(lib().aidax_pool_submit(self.h, x.ctypes.data_as(_fp), x.shape[1]))

    def register_host(self, arr: np.ndarray):
        """Page-lock a caller buffer once: blocks inside it are uploaded / downloaded without staging copies."""
        _check(lib().aidax_pool_register_host(self.h, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def unregister_host(self, arr: np.ndarray):
        _check(lib().aidax_pool_unregister_host(self.h, C.c_void_p(arr.ctypes.data)))

    def submit_to(self, x: np.ndarray, out: np.ndarray):
        assert x.dtype == np.float32 and out.dtype == np.float32 and x.flags.c_contiguous and out.flags.c_contiguous
        assert x.ndim == 2 and x.shape[0] == self.n_streams and out.size >= x.size
        _check(lib().aidax_pool_submit_to(self.h, x.ctypes.data_as(_fp), out.ctypes.data_as(_fp), x.shape[1]))

    def collect(self, n_frames: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.empty((self.n_streams, n_frames), np.float32)
        _check(lib().aidax_pool_collect(self.h, out.ctypes.data_as(_fp), n_frames))
        return out

    def process_device(self, d_in: int, d_out: int, n_frames: int, stream: int = 0):
        """d_in/d_out: raw device pointers ([n_streams][n_frames] fp32); stream: hipStream_t as int."""
        _check(lib().aidax_pool_process_device(self.h, C.c_void_p(d_in), C.c_void_p(d_out), n_frames,
                                               C.c_void_p(stream) if stream else None))

    def sync(self):
        _check(lib().aidax_pool_sync(self.h))

    def export_stream_dsp(self, stream: int) -> "StreamDsp":
        d = StreamDsp()
        _check(lib().aidax_pool_export_stream_dsp(self.h, stream, C.byref(d)))
        return d

    def import_stream_dsp(self, stream: int, d: "StreamDsp"):
        _check(lib().aidax_pool_import_stream_dsp(self.h, stream, C.byref(d)))

    def reset_stream(self, stream: int, start_mode: int = START_WARMUP):
        _check(lib().aidax_pool_reset_stream(self.h, stream, start_mode))

    def read_state(self, stream: int = 0, layer: int = 0, hidden: int = 128):
        h, c = np.zeros(hidden, np.float32), np.zeros(hidden, np.float32)
        H = _check(lib().aidax_pool_read_state(self.h, stream, layer, h.ctypes.data_as(_fp), c.ctypes.data_as(_fp), hidden))
        return h[:H], c[:H]

    @property
    def kernel_name(self) -> str:
        return lib().aidax_pool_kernel_name(self.h).decode()

    def close(self):
        if self.h:
            lib().aidax_pool_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Hub:
    """aidax_hub: many plugin instances of one process, one pool pass per audio period (one period of latency)."""

    def __init__(self, max_instances: int, max_frames: int = 256, samplerate: float = 48000.0, device: int = 0):
        h = C.c_void_p()
        _check(lib().aidax_hub_create(max_instances, max_frames, samplerate, device, C.byref(h)))
        self.h = h
        live_handles.add(self)

    def set_model(self, m: Optional[Model], start_mode: int = START_WARMUP):
        _check(lib().aidax_hub_set_model(self.h, m.h if m is not None else None, start_mode))

    def attach(self) -> int:
        slot = C.c_int32(-1)
        _check(lib().aidax_hub_attach(self.h, C.byref(slot)))
        return slot.value

    def detach(self, slot: int):
        _check(lib().aidax_hub_detach(self.h, slot))

    def attach_successor(self, prev: "Hub", prev_slot: int) -> int:
        slot = C.c_int32(-1)
        _check(lib().aidax_hub_attach_successor(self.h, prev.h if prev is not None else None, prev_slot, C.byref(slot)))
        return slot.value

    def adopt(self, slot: int, prev: "Hub", prev_slot: int):
        _check(lib().aidax_hub_adopt(self.h, slot, prev.h, prev_slot))

    def set_controls(self, slot: int, c: Controls):
        _check(lib().aidax_hub_set_controls(self.h, slot, C.byref(c)))

    def run(self, slot: int, x: np.ndarray) -> np.ndarray:
        x = _f32(x)
        out = np.empty_like(x)
        _check(lib().aidax_hub_run(self.h, slot, x.ctypes.data_as(_fp), out.ctypes.data_as(_fp), x.size))
        return out

    def set_deadline_us(self, us: int):
        _check(lib().aidax_hub_set_deadline_us(self.h, us))

    def flush(self):
        _check(lib().aidax_hub_flush(self.h))

    @property
    def deadline_launches(self) -> int:
        return lib().aidax_hub_deadline_launches(self.h)

    @property
    def latency_frames(self) -> int:
        return lib().aidax_hub_latency_frames(self.h)

    @property
    def attached(self) -> int:
        return lib().aidax_hub_attached(self.h)

    @property
    def launches(self) -> int:
        return lib().aidax_hub_launches(self.h)

    def close(self):
        if self.h:
            lib().aidax_hub_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
