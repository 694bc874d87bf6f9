import importlib, os, sys, glob
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle as O
ax = importlib.import_module("aidadsp-lv2_amd")
for p in sorted(glob.glob("tests/golden/models/*.json")):
    spec = O.load_model(p); m = ax.Model(p)
    pool = ax.Pool(2, 64); pool.set_model(m, ax.START_WARMUP)
    h, c = pool.read_state(1)
    oh, oc = O.OracleModel(spec, warmup=True).state()
    print(os.path.basename(p)[:30], "dh %.2e dc %.2e  max|c| %.3f" % (np.abs(h - oh).max(), np.abs(c - oc).max(), np.abs(oc).max()))
    pool.close()
