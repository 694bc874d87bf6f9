#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/. Run in the dev
container only (needs torch CPU; `--ref` also needs /root/reference):

    python tests/golden/make_golden.py            # torch-CPU NN goldens  -> nn_<case>.npz
    python tests/golden/make_golden.py --ref      # + reference DSP goldens -> dsp_ref.npz

What each fixture pins:
  * nn_<case>.npz — an INDEPENDENT implementation (torch.nn.LSTM/GRU/Conv1d on CPU,
    fp32) of the synthetic models of tests/modelgen.py. These cover what the
    reference's own goldens do not pin: GRU gate order / bias convention,
    conditioned inputs (input_size 2/3), hidden sizes != 12, stacked LSTM, conv1d.
  * dsp_ref.npz — outputs of the REFERENCE'S OWN common/Biquad.cpp and
    common/ValueSmoother.hpp (compiled into oracle/_ref by oracle/Makefile):
    filter designs, impulse/noise responses incl. a coefficient change
    mid-stream, smoother ramps incl. a target change mid-ramp.
The six bundled model files under tests/golden/models/ carry the reference's own
input_batch/output_batch vectors and need no generation.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import modelgen  # noqa: E402


def torch_forward(j, X):
    import torch
    torch.set_num_threads(1)
    x = torch.from_numpy(X.astype(np.float32))[None]          # [1][T][I]
    with torch.no_grad():
        for l in j["layers"]:
            t = l["type"]
            H = l["shape"][-1]
            w = [np.asarray(a, np.float32) for a in l["weights"]]
            if t == "lstm":
                m = torch.nn.LSTM(w[0].shape[0], H, batch_first=True)
                m.weight_ih_l0.copy_(torch.from_numpy(w[0].T.copy()))      # i,f,g,o == keras i,f,c,o
                m.weight_hh_l0.copy_(torch.from_numpy(w[1].T.copy()))
                m.bias_ih_l0.copy_(torch.from_numpy(w[2]))
                m.bias_hh_l0.zero_()
                x, _ = m(x)
            elif t == "gru":
                m = torch.nn.GRU(w[0].shape[0], H, batch_first=True)

                def zrh_to_rzn(a):          # keras z|r|h  ->  torch r|z|n along the last axis
                    return np.concatenate([a[..., H:2 * H], a[..., 0:H], a[..., 2 * H:3 * H]], axis=-1)
                m.weight_ih_l0.copy_(torch.from_numpy(zrh_to_rzn(w[0]).T.copy()))
                m.weight_hh_l0.copy_(torch.from_numpy(zrh_to_rzn(w[1]).T.copy()))
                m.bias_ih_l0.copy_(torch.from_numpy(zrh_to_rzn(w[2][0])))
                m.bias_hh_l0.copy_(torch.from_numpy(zrh_to_rzn(w[2][1])))
                x, _ = m(x)
            elif t == "dense":
                x = x @ torch.from_numpy(w[0]) + torch.from_numpy(w[1])
            elif t == "conv1d":
                K, d = l["kernel_size"][-1], l["dilation"][-1]
                ker = torch.from_numpy(np.transpose(w[0], (2, 1, 0)).copy())   # [k][in][out] -> [out][in][k]
                xp = torch.nn.functional.pad(x.transpose(1, 2), ((K - 1) * d, 0))
                x = torch.nn.functional.conv1d(xp, ker, torch.from_numpy(w[1]), dilation=d).transpose(1, 2)
            else:
                raise ValueError(t)
            act = l.get("activation", "")
            if t in ("dense", "conv1d") and act:
                x = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}[act](x)
    return x[0, :, 0].numpy().astype(np.float32)


def make_nn(force=False):
    for name, kw in modelgen.GOLDEN_CASES.items():
        if not force and os.path.exists(os.path.join(HERE, f"nn_{name}.npz")):
            continue                      # committed fixtures stay byte-stable; --force regenerates all
        j = modelgen.make_model(**kw)
        X = modelgen.golden_inputs(name, kw["input_size"])
        y = torch_forward(j, X)
        np.savez_compressed(os.path.join(HERE, f"nn_{name}.npz"), X=X, y=y)
        print(f"nn_{name}.npz  T={len(y)}  |y|max={np.abs(y).max():.4f}")


def make_dsp_ref():
    import ctypes as C
    from oracle import oracle as O
    R = O.ref_lib()
    if R is None:
        raise SystemExit("oracle/_ref not built (needs /root/reference)")
    fp = C.POINTER(C.c_float)
    rs = np.random.RandomState(7)
    noise = rs.uniform(-1, 1, 1024).astype(np.float32)
    imp = np.zeros(512, np.float32)
    imp[0] = 1.0
    fs = 48000.0
    lpf = lambda pc: float(O.lib().orc_lpf_fc(C.c_float(pc)))   # Fc values are inputs, stored in the fixture
    # (type, Fc, Q, gain) — every design instantiate()/applyToneControls() can produce, plus notch
    designs = [
        (0, lpf(66.216), 0.707, 0.0), (0, lpf(100.0), 0.707, 0.0), (0, lpf(12.5), 0.707, 0.0),
        (1, 35.0 / fs, 0.707, 0.0), (1, 35.0 / 44100.0, 0.707, 0.0),
        (2, 750.0 / fs, 0.707, 0.0), (2, 2000.0 / fs, 5.0, 3.0), (2, 150.0 / fs, 0.2, -8.0),
        (3, 1000.0 / fs, 1.0, 0.0),
        (4, 75.0 / fs, 0.707, 0.0), (4, 75.0 / fs, 0.707, 8.0), (4, 75.0 / fs, 0.707, -8.0),
        (4, 750.0 / fs, 1.2, -3.0), (4, 5000.0 / fs, 5.0, 8.0), (4, 150.0 / fs, 0.2, -8.0),
        (5, 305.0 / fs, 0.707, 0.0), (5, 305.0 / fs, 0.707, 4.0), (5, 75.0 / fs, 0.707, -8.0), (5, 600.0 / fs, 0.707, 8.0),
        (6, 2000.0 / fs, 0.707, 2.0), (6, 900.0 / fs, 0.707, 3.0), (6, 4000.0 / fs, 0.707, -8.0), (6, 1000.0 / fs, 0.707, 8.0),
    ]
    D = np.array(designs, np.float64)
    coeffs = np.zeros((len(designs), 5), np.float64)
    imp_out = np.zeros((len(designs), imp.size), np.float32)
    noise_out = np.zeros((len(designs), noise.size), np.float32)
    for i, (t, fc, q, g) in enumerate(designs):
        h = R.ref_biquad_new(int(t), fc, q, g)
        R.ref_biquad_coeffs(h, coeffs[i].ctypes.data_as(C.POINTER(C.c_double)))
        R.ref_biquad_block(h, imp_out[i].ctypes.data_as(fp), imp.ctypes.data_as(fp), imp.size)
        R.ref_biquad_free(h)
        h = R.ref_biquad_new(int(t), fc, q, g)
        R.ref_biquad_block(h, noise_out[i].ctypes.data_as(fp), noise.ctypes.data_as(fp), noise.size)
        R.ref_biquad_free(h)
    # coefficient change mid-stream without state reset (Biquad.cpp:60-65 keeps z1,z2)
    h = R.ref_biquad_new(4, 750.0 / fs, 0.707, 0.0)
    chg = np.zeros(noise.size, np.float32)
    R.ref_biquad_block(h, chg[:400].ctypes.data_as(fp), noise[:400].ctypes.data_as(fp), 400)
    R.ref_biquad_set(h, 2, 900.0 / fs, 2.5, 6.0)
    tail = np.zeros(noise.size - 400, np.float32)
    R.ref_biquad_block(h, tail.ctypes.data_as(fp), np.ascontiguousarray(noise[400:]).ctypes.data_as(fp), tail.size)
    chg[400:] = tail
    R.ref_biquad_free(h)

    def smoother(kind, sr, tc, script, n):
        """script: list of (at_sample, target); first entry initial (cleared) target."""
        h = getattr(R, f"ref_{kind}_new")(sr, tc, script[0][1])
        out = np.zeros(n, np.float32)
        pos = 0
        for at, tgt in script[1:] + [(n, None)]:
            seg = np.zeros(at - pos, np.float32)
            if seg.size:
                getattr(R, f"ref_{kind}_run")(h, seg.ctypes.data_as(fp), seg.size)
            out[pos:at] = seg
            pos = at
            if tgt is not None:
                getattr(R, f"ref_{kind}_set_target")(h, tgt)
        getattr(R, f"ref_{kind}_free")(h)
        return out

    exp_a = smoother("expsm", 48000.0, 0.1, [(0, 0.0), (0, 1.0)], 6000)                       # tests/src/test_smoothers.cpp
    exp_b = smoother("expsm", 44100.0, 0.1, [(0, 1.0), (0, 3.98107), (1000, 0.25), (1500, 0.0)], 4000)
    lin_a = smoother("linsm", 48000.0, 0.1, [(0, 0.0), (0, 1.0)], 6000)                       # tests/src/test_smoothers.cpp
    lin_b = smoother("linsm", 48000.0, 0.1, [(0, 0.2), (0, 0.9), (1000, 0.1), (1001, 0.1 + 1e-8), (3000, 0.5)], 9000)
    np.savez_compressed(os.path.join(HERE, "dsp_ref.npz"), designs=D, coeffs=coeffs, imp=imp, noise=noise,
                        imp_out=imp_out, noise_out=noise_out, chg_out=chg,
                        exp_a=exp_a, exp_b=exp_b, lin_a=lin_a, lin_b=lin_b)
    print("dsp_ref.npz", coeffs.shape, imp_out.shape)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", action="store_true")
    ap.add_argument("--no-nn", action="store_true")
    ap.add_argument("--force", action="store_true", help="regenerate nn_<case>.npz files that already exist")
    a = ap.parse_args()
    if not a.no_nn:
        make_nn(a.force)
    if a.ref:
        make_dsp_ref()
