"""tools/export_aidax.py: torch modules -> AIDA-X json -> the C oracle's loader and forward pass must give
torch's own output (documents gate reordering / bias conventions of the on-disk format)."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests import modelgen

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import export_aidax  # noqa: E402

torch = pytest.importorskip("torch")


def _torch_run(mods, X):
    x = torch.from_numpy(X)[None]
    with torch.no_grad():
        for m in mods:
            if isinstance(m, (torch.nn.LSTM, torch.nn.GRU)):
                x, _ = m(x)
            else:
                x = m(x)
    return x[0, :, 0].numpy()


@pytest.mark.parametrize("kind,hidden,isz,layers", [("lstm", 16, 1, 1), ("lstm", 32, 3, 1), ("gru", 8, 2, 1),
                                                    ("gru", 64, 3, 1), ("lstm", 24, 1, 2), ("gru", 12, 2, 3)])
def test_exported_model_reproduces_torch(kind, hidden, isz, layers, tmp_path):
    torch.manual_seed(hidden * 10 + isz)
    torch.set_num_threads(1)
    cls = torch.nn.LSTM if kind == "lstm" else torch.nn.GRU
    mods, cur = [], isz
    for _ in range(layers):
        mods.append(cls(cur, hidden, batch_first=True))
        cur = hidden
    mods.append(torch.nn.Linear(hidden, 1))
    path = str(tmp_path / "m.json")
    j = export_aidax.export(mods, isz, path)
    assert j["layers"][0]["type"] == kind and len(j["layers"]) == layers + 1
    spec = O.load_model(path)
    assert (spec.input_size, spec.input_skip) == (isz, 0)
    X = modelgen.golden_inputs(f"{kind}{hidden}", isz)[:1024]
    got = O.net_run(spec, X)
    want = _torch_run(mods, X)
    assert np.abs(got - want).max() < 2e-6, (kind, hidden, isz, np.abs(got - want).max())


def test_exported_file_loads_through_the_c_abi_loader(tmp_path):
    """Same file through the product's own json loader (no GPU needed for loading)."""
    import importlib
    ax = importlib.import_module("aidadsp-lv2_amd")
    torch.manual_seed(3)
    mods = [torch.nn.GRU(3, 40, batch_first=True), torch.nn.Linear(40, 1)]
    path = str(tmp_path / "g40.json")
    export_aidax.export(mods, 3, path, in_skip=1, in_gain_db=-6.0, out_gain_db=3.0)
    i = ax.Model(path).info
    assert (i.cell, i.hidden, i.input_size, i.n_rnn_layers, i.input_skip) == (1, 40, 3, 1, 1)
    assert abs(i.input_gain - 10 ** (-6.0 / 20)) < 1e-6 and abs(i.output_gain - 10 ** (3.0 / 20)) < 1e-6


@pytest.mark.parametrize("act", ["Tanh", "ReLU", "Sigmoid", None])
def test_exported_conv_stack_reproduces_torch_with_the_activation_that_follows(act, tmp_path):
    """torch's Conv1d has no activation of its own: the exporter takes it from the module that follows (none when
    nothing follows) and requires the causal padding convention of the on-disk format."""
    torch.manual_seed(11)
    torch.set_num_threads(1)
    C, k, n_layers = 8, 3, 4
    mods, torch_seq, cur = [], [], 1
    for l in range(n_layers):
        conv = torch.nn.Conv1d(cur, C, k, dilation=2 ** l, padding=0)
        mods.append(conv)
        if act:
            mods.append(getattr(torch.nn, act)())
        cur = C
    mods.append(torch.nn.Linear(C, 1))
    path = str(tmp_path / "c.json")
    j = export_aidax.export(mods, 1, path)
    assert [L["type"] for L in j["layers"]] == ["conv1d"] * n_layers + ["dense"]
    assert all(L["activation"] == (act.lower() if act else "") for L in j["layers"][:-1]) and j["layers"][-1]["activation"] == ""
    spec = O.load_model(path)
    X = modelgen.golden_inputs("convx", 1)[:1024]
    got = O.net_run(spec, X)
    # torch side: causal = left-pad every conv's input by (k-1)*dilation
    x = torch.from_numpy(X.T.copy())[None]                 # [1][C=1][T]
    with torch.no_grad():
        for m in mods:
            if isinstance(m, torch.nn.Conv1d):
                x = m(torch.nn.functional.pad(x, ((m.kernel_size[0] - 1) * m.dilation[0], 0)))
            elif isinstance(m, torch.nn.Linear):
                x = m(x.transpose(1, 2)).transpose(1, 2)
            else:
                x = m(x)
    want = x[0, 0].numpy()
    assert np.abs(got - want).max() < 2e-6, (act, np.abs(got - want).max())


def test_exporter_rejects_misplaced_activations_and_non_causal_padding(tmp_path):
    with pytest.raises(TypeError):
        export_aidax.export([torch.nn.LSTM(1, 8, batch_first=True), torch.nn.Tanh(), torch.nn.Linear(8, 1)], 1)
    with pytest.raises(TypeError):
        export_aidax.export([torch.nn.Tanh(), torch.nn.Linear(1, 1)], 1)
    with pytest.raises(AssertionError):
        export_aidax.export([torch.nn.Conv1d(1, 4, 3, padding=1), torch.nn.Linear(4, 1)], 1)     # "same" padding: not causal
