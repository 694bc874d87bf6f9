"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker. The product path is the HIP library
behind ``include/aidax.h``; it never routes through here.

The json handling mirrors the reference loader's key semantics
(rt-neural-generic/src/rt-neural-generic.cpp:977-1013 and the architecture
predicate model_variant.hpp:62-71): ``in_shape[-1]`` = input size (<=3),
``in_skip`` only if a number (must be 0/1), ``in_gain``/``out_gain`` dB->linear
only if numbers, sample rate only if ``metadata.samplerate`` / ``samplerate`` is
a *number* (the bundled files carry the string "48000", so 48000.0 applies).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libaidax_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libaidadsp_ref.so")

L_LSTM, L_GRU, L_DENSE, L_CONV1D = 0, 1, 2, 3
ACT = {"": 0, None: 0, "linear": 0, "tanh": 1, "relu": 2, "sigmoid": 3}
BQ_LOWPASS, BQ_HIGHPASS, BQ_BANDPASS, BQ_NOTCH, BQ_PEAK, BQ_LOWSHELF, BQ_HIGHSHELF = range(7)

_fp = C.POINTER(C.c_float)


class LayerDesc(C.Structure):
    _fields_ = [("type", C.c_int), ("in_size", C.c_int), ("out_size", C.c_int),
                ("activation", C.c_int), ("ksize", C.c_int), ("dilation", C.c_int),
                ("w0", _fp), ("w1", _fp), ("w2", _fp)]


class Biquad(C.Structure):
    _fields_ = [("type", C.c_int)] + [(n, C.c_double) for n in
                ("a0", "a1", "a2", "b1", "b2", "Fc", "Q", "peakGain", "z1", "z2")]


class ExpSm(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("coef", "target", "mem", "tau", "sampleRate")]


class LinSm(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("step", "target", "mem", "tau", "sampleRate")]


CONTROL_FIELDS = ("in_lpf_pc", "pregain_db", "net_bypass", "param1", "param2", "eq_bypass",
                  "eq_position", "bass_boost_db", "bass_freq", "mid_boost_db", "mid_freq", "mid_q",
                  "mid_type", "treble_boost_db", "treble_freq", "depth_boost_db",
                  "presence_boost_db", "dc_blocker", "master_db", "enabled")


class Controls(C.Structure):
    _fields_ = [(n, C.c_float) for n in CONTROL_FIELDS]


class DynModel(C.Structure):
    _fields_ = [("net", C.c_void_p), ("input_size", C.c_int), ("input_skip", C.c_int),
                ("input_gain", C.c_float), ("output_gain", C.c_float), ("samplerate", C.c_float),
                ("param1Coeff", LinSm), ("param2Coeff", LinSm), ("paramFirstRun", C.c_int)]


class Plugin(C.Structure):
    _fields_ = ([("samplerate", C.c_double), ("preGain", ExpSm), ("masterGain", ExpSm)] +
                [(n, Biquad) for n in ("dc_blocker", "in_lpf", "bass", "mid", "treble", "depth", "presence")] +
                [(n, C.c_float) for n in ("in_lpf_pc_old", "bass_boost_db_old", "bass_freq_old",
                                          "mid_boost_db_old", "mid_freq_old", "mid_q_old", "mid_type_old",
                                          "treble_boost_db_old", "treble_freq_old",
                                          "depth_boost_db_old", "presence_boost_db_old")] +
                [("loading", C.c_int), ("model", C.POINTER(DynModel))])


def build(force: bool = False) -> None:
    """Compile the C restatement (and _ref where the reference tree exists)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.isdir("/root/reference/common") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.orc_biquad_init.argtypes = [C.POINTER(Biquad), C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_biquad_set.argtypes = [C.POINTER(Biquad), C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_biquad_block.argtypes = [C.POINTER(Biquad), _fp, _fp, C.c_uint32]
    L.orc_biquad_process.argtypes = [C.POINTER(Biquad), C.c_float]
    L.orc_biquad_process.restype = C.c_float
    for n in ("orc_expsm_init", "orc_expsm_clear_to_target"):
        getattr(L, n).argtypes = [C.POINTER(ExpSm)]
    for n in ("orc_expsm_set_sample_rate", "orc_expsm_set_time_constant", "orc_expsm_set_target"):
        getattr(L, n).argtypes = [C.POINTER(ExpSm), C.c_float]
    L.orc_expsm_next.argtypes = [C.POINTER(ExpSm)]
    L.orc_expsm_next.restype = C.c_float
    for n in ("orc_linsm_init", "orc_linsm_clear_to_target"):
        getattr(L, n).argtypes = [C.POINTER(LinSm)]
    for n in ("orc_linsm_set_sample_rate", "orc_linsm_set_time_constant", "orc_linsm_set_target"):
        getattr(L, n).argtypes = [C.POINTER(LinSm), C.c_float]
    L.orc_linsm_next.argtypes = [C.POINTER(LinSm)]
    L.orc_linsm_next.restype = C.c_float
    L.orc_db_co.argtypes = [C.c_float]
    L.orc_db_co.restype = C.c_float
    L.orc_lpf_fc.argtypes = [C.c_float]
    L.orc_lpf_fc.restype = C.c_float
    L.orc_net_create.argtypes = [C.POINTER(LayerDesc), C.c_int, C.c_int]
    L.orc_net_create.restype = C.c_void_p
    L.orc_net_free.argtypes = [C.c_void_p]
    L.orc_net_reset.argtypes = [C.c_void_p]
    L.orc_net_forward.argtypes = [C.c_void_p, _fp]
    L.orc_net_forward.restype = C.c_float
    L.orc_net_state.argtypes = [C.c_void_p, C.c_int, _fp, _fp, C.c_int]
    L.orc_net_state.restype = C.c_int
    L.orc_dynmodel_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_int]
    L.orc_dynmodel_create.restype = C.POINTER(DynModel)
    L.orc_dynmodel_free.argtypes = [C.POINTER(DynModel)]
    L.orc_apply_model.argtypes = [C.POINTER(DynModel), _fp, C.c_uint32]
    L.orc_test_model.argtypes = [C.POINTER(DynModel), _fp, _fp, C.c_uint32, C.c_double, _fp, _fp]
    L.orc_test_model.restype = C.c_int
    L.orc_controls_default.argtypes = [C.POINTER(Controls)]
    L.orc_plugin_init.argtypes = [C.POINTER(Plugin), C.c_double]
    L.orc_plugin_activate.argtypes = [C.POINTER(Plugin)]
    L.orc_plugin_set_model.argtypes = [C.POINTER(Plugin), C.POINTER(DynModel)]
    L.orc_plugin_run.argtypes = [C.POINTER(Plugin), C.POINTER(Controls), _fp, _fp, C.c_uint32]
    L.orc_bench.argtypes = [C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                            C.POINTER(Controls), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int]
    L.orc_bench.restype = C.c_double
    _lib = L
    return L


def ref_lib() -> Optional[C.CDLL]:
    """The reference's own Biquad/ValueSmoother build, or None when unavailable."""
    build()
    if not os.path.exists(_REF_PATH):
        return None
    R = C.CDLL(_REF_PATH)
    R.ref_biquad_new.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    R.ref_biquad_new.restype = C.c_void_p
    R.ref_biquad_free.argtypes = [C.c_void_p]
    R.ref_biquad_set.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    R.ref_biquad_coeffs.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    R.ref_biquad_state.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    R.ref_biquad_block.argtypes = [C.c_void_p, _fp, _fp, C.c_uint32]
    for k in ("expsm", "linsm"):
        getattr(R, f"ref_{k}_new").argtypes = [C.c_float, C.c_float, C.c_float]
        getattr(R, f"ref_{k}_new").restype = C.c_void_p
        getattr(R, f"ref_{k}_free").argtypes = [C.c_void_p]
        getattr(R, f"ref_{k}_set_target").argtypes = [C.c_void_p, C.c_float]
        getattr(R, f"ref_{k}_clear").argtypes = [C.c_void_p]
        getattr(R, f"ref_{k}_run").argtypes = [C.c_void_p, _fp, C.c_uint32]
    return R


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_fp)


def db_co(db: float) -> float:
    return float(lib().orc_db_co(C.c_float(db)))


@dataclass
class ModelSpec:
    """An AIDA-X json model, parsed the way the reference loader reads it."""
    layers: List[dict]
    input_size: int
    input_skip: int
    input_gain: float
    output_gain: float
    samplerate: float
    input_batch: Optional[np.ndarray] = None
    output_batch: Optional[np.ndarray] = None
    arrays: list = field(default_factory=list)   # keep numpy weight buffers alive

    @property
    def rnn_type(self) -> str:
        return self.layers[0]["type"]

    @property
    def hidden(self) -> int:
        return int(self.layers[0]["shape"][-1])

    def descs(self):
        """(LayerDesc array, count). The array points into numpy buffers that must outlive the C call that reads them:
        they hang on the returned array itself (`arr.keep`), so two threads building models from one spec do not free
        each other's."""
        n = len(self.layers)
        arr = (LayerDesc * n)()
        arrays = []
        arr.keep = arrays
        self.arrays = arrays
        cur_in = self.input_size
        for i, l in enumerate(self.layers):
            t = l["type"]
            out = int(l["shape"][-1])
            w = [_f32(x) for x in l["weights"]]
            arrays.extend(w)
            d = arr[i]
            d.in_size, d.out_size = cur_in, out
            d.activation = ACT[l.get("activation", "")]
            d.ksize, d.dilation = 0, 0
            if t == "lstm":
                d.type = L_LSTM
                assert w[0].shape == (cur_in, 4 * out) and w[1].shape == (out, 4 * out) and w[2].shape == (4 * out,)
                d.w0, d.w1, d.w2 = _ptr(w[0]), _ptr(w[1]), _ptr(w[2])
            elif t == "gru":
                d.type = L_GRU
                assert w[0].shape == (cur_in, 3 * out) and w[1].shape == (out, 3 * out) and w[2].shape == (2, 3 * out)
                d.w0, d.w1, d.w2 = _ptr(w[0]), _ptr(w[1]), _ptr(w[2])
            elif t == "dense":
                d.type = L_DENSE
                assert w[0].shape == (cur_in, out) and w[1].shape == (out,)
                d.w0, d.w1 = _ptr(w[0]), _ptr(w[1])
            elif t == "conv1d":
                d.type = L_CONV1D
                d.ksize = int(np.atleast_1d(l["kernel_size"])[-1])
                d.dilation = int(np.atleast_1d(l["dilation"])[-1])
                assert w[0].shape == (d.ksize, cur_in, out) and w[1].shape == (out,)
                d.w0, d.w1 = _ptr(w[0]), _ptr(w[1])
            else:
                raise ValueError(f"unsupported layer type {t!r}")
            cur_in = out
        return arr, n


def _is_number(v) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def parse_model(j: dict) -> ModelSpec:
    """Key semantics of loadModelFromPath (rt-neural-generic.cpp:977-1013)."""
    input_size = int(j["in_shape"][-1])
    if input_size > 3:
        raise ValueError("Value for input_size not supported")
    skip = 0
    if _is_number(j.get("in_skip")):
        skip = int(j["in_skip"])
        if skip > 1:
            raise ValueError("Values for in_skip > 1 are not supported")
    ig = db_co(float(j["in_gain"])) if _is_number(j.get("in_gain")) else 1.0
    og = db_co(float(j["out_gain"])) if _is_number(j.get("out_gain")) else 1.0
    md = j.get("metadata") or {}
    if _is_number(md.get("samplerate")):
        sr = float(md["samplerate"])
    elif _is_number(j.get("samplerate")):
        sr = float(j["samplerate"])
    else:
        sr = 48000.0
    spec = ModelSpec(layers=j["layers"], input_size=input_size, input_skip=skip,
                     input_gain=ig, output_gain=og, samplerate=sr)
    if isinstance(j.get("input_batch"), list) and isinstance(j.get("output_batch"), list):
        spec.input_batch = _f32(j["input_batch"])
        spec.output_batch = _f32(j["output_batch"])
    return spec


def load_model(path: str) -> ModelSpec:
    with open(path, "rb") as f:
        return parse_model(json.load(f))


class OracleModel:
    """DynamicModel mirror (rt-neural-generic.h:115-129)."""

    def __init__(self, spec: ModelSpec, old_param1: float = 0.0, old_param2: float = 0.0,
                 warmup: bool = True, f64: bool = False):
        self.spec = spec
        L = lib()
        arr, n = spec.descs()
        self._descs = arr
        net = L.orc_net_create(arr, n, 1 if f64 else 0)
        self.ptr = L.orc_dynmodel_create(net, spec.input_size, spec.input_skip,
                                         spec.input_gain, spec.output_gain, spec.samplerate,
                                         old_param1, old_param2, 1 if warmup else 0)
        self._owned = True

    def apply(self, buf: np.ndarray) -> np.ndarray:
        out = _f32(buf).copy()
        lib().orc_apply_model(self.ptr, _ptr(out), out.size)
        return out

    def test_model(self, x: np.ndarray, y: np.ndarray, thr: float = 1.0e-5):
        x, y = _f32(x), _f32(y)
        out = np.empty_like(x)
        me = C.c_float(0)
        n_err = lib().orc_test_model(self.ptr, _ptr(x), _ptr(y), x.size, thr, C.byref(me), _ptr(out))
        return n_err, float(me.value), out

    def state(self, layer: int = 0):
        H = int(self.spec.layers[layer]["shape"][-1])
        h = np.zeros(H, np.float32)
        c = np.zeros(H, np.float32)
        lib().orc_net_state(self.ptr.contents.net, layer, _ptr(h), _ptr(c), H)
        return h, c

    def release(self):
        self._owned = False

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            lib().orc_dynmodel_free(self.ptr)
            self.ptr = None


def net_run(spec: ModelSpec, X: np.ndarray, f64: bool = False, flavour: Optional[int] = None) -> np.ndarray:
    """Bare network (reset state) over a [T][input_size] sequence -> [T] outputs."""
    X = _f32(X).reshape(-1, spec.input_size)
    L = lib()
    arr, n = spec.descs()
    net = L.orc_net_create(arr, n, flavour if flavour is not None else (1 if f64 else 0))
    y = np.empty(X.shape[0], np.float32)
    try:
        for t in range(X.shape[0]):
            y[t] = L.orc_net_forward(net, _ptr(X[t]))
    finally:
        L.orc_net_free(net)
    return y


def default_controls(**kw) -> Controls:
    c = Controls()
    lib().orc_controls_default(C.byref(c))
    for k, v in kw.items():
        if k not in CONTROL_FIELDS:
            raise KeyError(k)
        setattr(c, k, v)
    return c


class OraclePlugin:
    """One plugin instance = one mono stream (RtNeuralGeneric DSP members)."""

    def __init__(self, samplerate: float = 48000.0):
        self.p = Plugin()
        lib().orc_plugin_init(C.byref(self.p), samplerate)
        self.model: Optional[OracleModel] = None

    def set_model(self, m: OracleModel):
        self.model = m
        lib().orc_plugin_set_model(C.byref(self.p), m.ptr)

    def activate(self):
        lib().orc_plugin_activate(C.byref(self.p))

    def set_loading(self, v: bool):
        self.p.loading = 1 if v else 0

    def run(self, c: Controls, x: np.ndarray) -> np.ndarray:
        x = _f32(x)
        out = np.empty_like(x)
        lib().orc_plugin_run(C.byref(self.p), C.byref(c), _ptr(x), _ptr(out), x.size)
        return out


def run_streams(spec: Optional[ModelSpec], controls: Sequence[Controls] | Controls, x: np.ndarray,
                block: int, samplerate: float = 48000.0, warmup: bool = True,
                old_params=(0.0, 0.0), loading: bool = False) -> np.ndarray:
    """Oracle for the batched boundary: x is [n_streams][n_total]; processed in
    `block`-frame run() calls per stream with per-stream state carried across."""
    x = _f32(x)
    S, N = x.shape
    out = np.empty_like(x)
    for s in range(S):
        pl = OraclePlugin(samplerate)
        if spec is not None:
            pl.set_model(OracleModel(spec, old_params[0], old_params[1], warmup=warmup))
        pl.set_loading(loading or spec is None)
        c = controls if isinstance(controls, Controls) else controls[s]
        for b in range(0, N, block):
            out[s, b:b + block] = pl.run(c, x[s, b:b + block])
    return out


def cpu_bench(spec: ModelSpec, controls: Controls, x: np.ndarray, n_blocks: int, warm_blocks: int,
              n_threads: int, fast: bool = False):
    """Timed multi-thread pass of the full chain; returns (seconds, last-block output)."""
    x = _f32(x)
    S, F = x.shape
    arr, n = spec.descs()
    out = np.empty_like(x)
    secs = lib().orc_bench(arr, n, spec.input_size, spec.input_skip, spec.input_gain, spec.output_gain,
                           C.byref(controls), S, F, n_blocks, warm_blocks, n_threads, _ptr(x), _ptr(out), 2 if fast else 0)
    return float(secs), out
