"""Deterministic synthetic AIDA-X model files and input signals: the workloads of bench.py (SURVEY §8(d):
synthetic weights and inputs, there are no checkpoints or datasets offline) and of the parity tests
(tests/modelgen.py re-exports this module).

The json schema written here is the one the reference loader reads
(rt-neural-generic/src/rt-neural-generic.cpp:977-1013; architecture predicate
model_variant.hpp:62-71): ``in_shape``, optional ``in_skip``/``in_gain``/``out_gain``,
``layers[i] = {type, activation, shape, weights}`` with Keras weight layouts
(LSTM ``[I][4H] | [H][4H] | [4H]`` columns i|f|c|o; GRU ``[I][3H] | [H][3H] | [2][3H]``
columns z|r|h; Dense ``[H][1] | [1]``). ``conv1d`` layers use RTNeural's published
json keys (``kernel_size``, ``dilation``, kernel ``[k][in][out]``) and, like stacked
LSTM layers, are an extension the reference itself cannot load.

Only numpy's legacy ``RandomState`` is used, whose streams are frozen across
versions, so the same (case, seed) always yields the same model and signal.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np

HIDDEN_SIZES = (8, 12, 16, 20, 24, 32, 40, 64, 80)     # variant/generate_variant_hpp.py:6
INPUT_SIZES = (1, 2, 3)                                # variant/generate_variant_hpp.py:5


def _u(rs, shape, scale):
    return rs.uniform(-scale, scale, size=shape).astype(np.float32)


def rnn_layer(rs, kind: str, in_size: int, hidden: int) -> dict:
    g = 4 if kind == "lstm" else 3
    k = 1.0 / np.sqrt(hidden)
    W = _u(rs, (in_size, g * hidden), 2.0 * k)          # input weights a bit hot: full-scale audio
    U = _u(rs, (hidden, g * hidden), k)                 # should reach the nonlinear regime
    if kind == "lstm":
        b = _u(rs, (g * hidden,), k)
        b[hidden:2 * hidden] += 1.0                     # forget-gate bias, as trained models have
    else:
        b = _u(rs, (2, g * hidden), k)
    return {"type": kind, "activation": "", "shape": [None, None, hidden],
            "weights": [W.tolist(), U.tolist(), b.tolist()]}


def dense_layer(rs, in_size: int, out: int = 1, activation: str = "") -> dict:
    k = 1.0 / np.sqrt(in_size)
    return {"type": "dense", "activation": activation, "shape": [None, None, out],
            "weights": [_u(rs, (in_size, out), k).tolist(), _u(rs, (out,), k).tolist()]}


def conv_layer(rs, in_size: int, out: int, ksize: int, dilation: int, activation: str = "tanh") -> dict:
    k = 1.0 / np.sqrt(in_size * ksize)
    return {"type": "conv1d", "activation": activation, "shape": [None, None, out],
            "kernel_size": [ksize], "dilation": [dilation],
            "weights": [_u(rs, (ksize, in_size, out), 1.5 * k).tolist(), _u(rs, (out,), k).tolist()]}


def make_model(kind: str, hidden: int, input_size: int = 1, seed: int = 0, n_rnn: int = 1,
               in_skip: Optional[int] = None, in_gain: Optional[float] = None,
               out_gain: Optional[float] = None, samplerate=None,
               conv_layers: int = 8, conv_k: int = 3, conv_dilations=None) -> Dict:
    """kind in {'lstm','gru','conv'}; returns the json dict. conv: conv_layers layers of conv_k taps, layer l dilated by 2^l — or by
    conv_dilations[l] (then as many layers as it has entries)."""
    rs = np.random.RandomState(seed)
    layers = []
    if kind in ("lstm", "gru"):
        cur = input_size
        for _ in range(n_rnn):
            layers.append(rnn_layer(rs, kind, cur, hidden))
            cur = hidden
        layers.append(dense_layer(rs, hidden))
    elif kind == "conv":
        cur = input_size
        dil = list(conv_dilations) if conv_dilations is not None else [2 ** l for l in range(conv_layers)]
        for l in range(len(dil)):
            layers.append(conv_layer(rs, cur, hidden, conv_k, dil[l]))
            cur = hidden
        layers.append(dense_layer(rs, hidden))
    else:
        raise ValueError(kind)
    j = {"in_shape": [None, None, input_size], "layers": layers,
         "metadata": {"name": f"synthetic_{kind}{hidden}x{n_rnn}_in{input_size}_seed{seed}",
                      "samplerate": "48000"}}
    if in_skip is not None:
        j["in_skip"] = in_skip
    if in_gain is not None:
        j["in_gain"] = in_gain
    if out_gain is not None:
        j["out_gain"] = out_gain
    if samplerate is not None:
        j["samplerate"] = samplerate
    return j


def write_model(j: Dict, path: str) -> str:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        json.dump(j, f)
    return path


def signal(n_streams: int, n: int, seed: int = 0xA1DA, fs: float = 48000.0) -> np.ndarray:
    """SURVEY §8(d) synthetic input: uniform noise in [-0.5,0.5]*0.5 mixed with a
    110*2^(s mod 5) Hz sine at 0.25 — guitar-level, exercises the nonlinearity."""
    out = np.empty((n_streams, n), np.float32)
    t = np.arange(n, dtype=np.float64) / fs
    for s in range(n_streams):
        rs = np.random.RandomState((seed ^ s) & 0x7FFFFFFF)
        noise = rs.uniform(-0.5, 0.5, size=n)
        sine = 0.25 * np.sin(2.0 * np.pi * 110.0 * (2 ** (s % 5)) * t)
        out[s] = (0.5 * noise + sine).astype(np.float32)
    return out


# Named golden cases: tests/golden/make_golden.py writes <name>.npz for each.
GOLDEN_CASES = {
    #  name                kwargs for make_model
    "lstm16_in1":          dict(kind="lstm", hidden=16, input_size=1, seed=16),
    "lstm32_in1":          dict(kind="lstm", hidden=32, input_size=1, seed=32),
    "lstm32_in1_skip":     dict(kind="lstm", hidden=32, input_size=1, seed=33, in_skip=1, in_gain=-3.0, out_gain=4.5),
    "lstm12_in3":          dict(kind="lstm", hidden=12, input_size=3, seed=12),
    "lstm80_in2":          dict(kind="lstm", hidden=80, input_size=2, seed=80),
    "gru8_in1":            dict(kind="gru", hidden=8, input_size=1, seed=8),
    "gru32_in2":           dict(kind="gru", hidden=32, input_size=2, seed=322),
    "gru64_in3":           dict(kind="gru", hidden=64, input_size=3, seed=64),
    "gru80_in1":           dict(kind="gru", hidden=80, input_size=1, seed=801),
    "lstm96x2_in1":        dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2),
    "conv16x8_in1":        dict(kind="conv", hidden=16, input_size=1, seed=1608),
    # widths outside the reference's table: served by the matrix-core kernel (k_mfma)
    "gru128_in3":          dict(kind="gru", hidden=128, input_size=3, seed=128),
    "gru48x3_in2":         dict(kind="gru", hidden=48, input_size=2, seed=483, n_rnn=3),
    "lstm112_in1_skip":    dict(kind="lstm", hidden=112, input_size=1, seed=112, in_skip=1),
}
GOLDEN_LEN = 4096


def golden_inputs(name: str, input_size: int) -> np.ndarray:
    """[T][input_size]: audio in column 0, slow 0..1 ramps in the param columns."""
    seed = sum(ord(c) for c in name)
    x = signal(1, GOLDEN_LEN, seed=seed)[0]
    X = np.zeros((GOLDEN_LEN, input_size), np.float32)
    X[:, 0] = x
    t = np.arange(GOLDEN_LEN, dtype=np.float32) / GOLDEN_LEN
    if input_size >= 2:
        X[:, 1] = t                      # 0 -> 1
    if input_size >= 3:
        X[:, 2] = 1.0 - 0.7 * t          # 1 -> 0.3
    return X
