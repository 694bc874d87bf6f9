import importlib, json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
for cell, H, I in [("gru", 8, 1), ("gru", 64, 3), ("lstm", 64, 1), ("lstm", 32, 1), ("lstm", 12, 1), ("lstm", 16, 2), ("lstm", 40, 1), ("lstm",80,1), ("gru",80,1)]:
    j = modelgen.make_model(cell, H, I, seed=H)
    p = modelgen.write_model(j, os.path.join(d, f"{cell}{H}.json"))
    spec = O.parse_model(j)
    X = modelgen.golden_inputs("x", I)[:64]
    y = ax.Model(p).forward(X, unit_gains=True)
    r = O.net_run(spec, X)
    print(cell, H, I, "maxerr", np.abs(y - r).max(), "gpu", y[:4], "ref", r[:4])
