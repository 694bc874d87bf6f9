#!/usr/bin/env python3
"""torch -> AIDA-X .json exporter.

Writes the schema the reference loader reads (rt-neural-generic/src/rt-neural-generic.cpp:977-1013):

    {"in_shape": [null, null, I], "in_skip": 0|1, "in_gain": dB, "out_gain": dB,
     "metadata": {"samplerate": ...}, "layers": [{type, activation, shape, weights}, ...]}

with the Keras weight layouts RTNeural's parseJson expects:

    torch.nn.LSTM  weight_ih [4H][I] rows i|f|g|o  ->  W [I][4H] cols i|f|c|o   (same gate order, transposed)
                   weight_hh [4H][H]               ->  U [H][4H]
                   bias_ih + bias_hh               ->  b [4H]                    (Keras LSTM has one bias)
    torch.nn.GRU   weight_ih [3H][I] rows r|z|n    ->  W [I][3H] cols z|r|h      (gate blocks reordered)
                   weight_hh [3H][H]               ->  U [H][3H]
                   bias_ih, bias_hh                ->  b [2][3H]                 (reset_after: both kept)
    torch.nn.Linear weight [1][H], bias [1]        ->  W [H][1], b [1]
    torch.nn.Conv1d weight [out][in][k]            ->  K [k][in][out], keys kernel_size / dilation (extension)

Usage from Python: ``export(modules, input_size, path, in_skip=0, ...)`` where modules is a list of
torch.nn.LSTM / GRU / Conv1d layers (single-layer, batch_first irrelevant) followed by one torch.nn.Linear.
"""
from __future__ import annotations

import json
from typing import Optional, Sequence

import numpy as np


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _rzn_to_zrh(a: np.ndarray, H: int) -> np.ndarray:
    """torch GRU blocks r|z|n along the LAST axis -> keras z|r|h."""
    return np.concatenate([a[..., H:2 * H], a[..., 0:H], a[..., 2 * H:3 * H]], axis=-1)


_ACTIVATIONS = {"Tanh": "tanh", "ReLU": "relu", "Sigmoid": "sigmoid"}


def layer_dict(m, activation: str = "") -> dict:
    """One weight-carrying module -> one json layer. `activation` ("", "tanh", "relu", "sigmoid") is what FOLLOWS the
    module in the network (export() takes it from the next module of the sequence); torch's Conv1d / Linear have
    none of their own. The recurrent layers carry their gate activations implicitly."""
    import torch
    if isinstance(m, torch.nn.LSTM):
        assert m.num_layers == 1 and not m.bidirectional and m.proj_size == 0
        H = m.hidden_size
        b = _np(m.bias_ih_l0) + _np(m.bias_hh_l0) if m.bias else np.zeros(4 * H, np.float32)
        return {"type": "lstm", "activation": "", "shape": [None, None, H],
                "weights": [_np(m.weight_ih_l0).T.tolist(), _np(m.weight_hh_l0).T.tolist(), b.tolist()]}
    if isinstance(m, torch.nn.GRU):
        assert m.num_layers == 1 and not m.bidirectional
        H = m.hidden_size
        W = _rzn_to_zrh(_np(m.weight_ih_l0).T, H)
        U = _rzn_to_zrh(_np(m.weight_hh_l0).T, H)
        if m.bias:
            b = np.stack([_rzn_to_zrh(_np(m.bias_ih_l0), H), _rzn_to_zrh(_np(m.bias_hh_l0), H)])
        else:
            b = np.zeros((2, 3 * H), np.float32)
        return {"type": "gru", "activation": "", "shape": [None, None, H],
                "weights": [W.tolist(), U.tolist(), b.tolist()]}
    if isinstance(m, torch.nn.Linear):
        b = _np(m.bias) if m.bias is not None else np.zeros(m.out_features, np.float32)
        return {"type": "dense", "activation": activation, "shape": [None, None, m.out_features],
                "weights": [_np(m.weight).T.tolist(), b.tolist()]}
    if isinstance(m, torch.nn.Conv1d):
        # The AIDA-X / RTNeural conv1d is CAUSAL: output frame t sees frames t, t-d, ..., t-(k-1)d. A torch Conv1d
        # computes that when the caller left-pads its input by (k-1)*d (padding=0 on the module, F.pad before it) or
        # uses padding=(k-1)*d and drops the last (k-1)*d outputs; "same"-style symmetric padding is a different filter.
        assert m.stride == (1,) and m.groups == 1
        assert m.padding_mode == "zeros" and m.padding in ((0,), ((m.kernel_size[0] - 1) * m.dilation[0],)), \
            "conv1d layers are causal: pad the input on the left by (k-1)*dilation (see the comment above)"
        b = _np(m.bias) if m.bias is not None else np.zeros(m.out_channels, np.float32)
        return {"type": "conv1d", "activation": activation, "shape": [None, None, m.out_channels],
                "kernel_size": [m.kernel_size[0]], "dilation": [m.dilation[0]],
                "weights": [np.transpose(_np(m.weight), (2, 1, 0)).tolist(), b.tolist()]}
    raise TypeError(f"no AIDA-X layer for {type(m).__name__}")


def export(modules: Sequence, input_size: int, path: Optional[str] = None, in_skip: int = 0,
           in_gain_db: Optional[float] = None, out_gain_db: Optional[float] = None,
           samplerate: float = 48000.0, name: str = "exported") -> dict:
    """`modules`: the network in order. nn.Tanh / nn.ReLU / nn.Sigmoid modules are not layers of their own in the
    AIDA-X schema: each becomes the "activation" of the Conv1d / Linear in front of it (a bare activation anywhere
    else is an error, as is one after an LSTM / GRU)."""
    import torch
    layers = []
    mods = list(modules)
    i = 0
    while i < len(mods):
        m = mods[i]
        act = ""
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if type(m).__name__ in _ACTIVATIONS:
            raise TypeError(f"{type(m).__name__} at position {i} does not follow a Conv1d or Linear")
        if nxt is not None and type(nxt).__name__ in _ACTIVATIONS:
            if not isinstance(m, (torch.nn.Conv1d, torch.nn.Linear)):
                raise TypeError(f"{type(nxt).__name__} after {type(m).__name__}: only Conv1d / Linear take an activation")
            act = _ACTIVATIONS[type(nxt).__name__]
            i += 1
        layers.append(layer_dict(m, act))
        i += 1
    j = {"in_shape": [None, None, int(input_size)], "in_skip": int(in_skip),
         "layers": layers,
         "metadata": {"name": name, "samplerate": samplerate}}
    if in_gain_db is not None:
        j["in_gain"] = float(in_gain_db)
    if out_gain_db is not None:
        j["out_gain"] = float(out_gain_db)
    if path:
        with open(path, "w") as f:
            json.dump(j, f)
    return j
