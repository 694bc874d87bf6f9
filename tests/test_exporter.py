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
