#!/bin/bash
# A/B: tanh(c) of the LSTM cell in the exp form (5 instructions) against the rational (15): error on the bundled goldens,
# 48000-sample drift, cfg2 time. Builds the library twice in place; leaves the default build behind.
set -e
cd "$(dirname "$0")/.."
for flag in "-DAIDAX_TANHC_EXP" ""; do
    touch aidadsp-lv2_amd/csrc/aidax_kernels.hip
    make -s -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wextra -Iinclude -fno-slp-vectorize $flag" > /dev/null
    echo "=== build with [$flag]"
    python - <<'PY'
import importlib, os, sys, glob
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
worst = 0
for f in sorted(glob.glob("tests/golden/models/*.json")):
    m = ax.Model(f); n_err, mx = m.self_test()[:2]
    worst = max(worst, mx or 0); print(os.path.basename(f)[:28], "self-test max err", mx)
# drift: one second of full-scale noise through LSTM-32 and the hottest bundled model
for name, path in (("lstm32", modelgen.write_model(modelgen.make_model("lstm", 32, 1, seed=32), "build/l32.json")), ("british_lead", [g for g in glob.glob("tests/golden/models/*british*")][0])):
    spec = O.load_model(path); pool = ax.Pool(2, 256); pool.set_model(ax.Model(path)); pool.set_controls(ax.default_controls())
    x = modelgen.signal(2, 48128, seed=5); got = np.concatenate([pool.process(np.ascontiguousarray(x[:, b:b+256])) for b in range(0, 48128, 256)], axis=1)
    want = O.run_streams(spec, O.default_controls(), x, 256)
    print(name, "48128-sample drift max err", float(np.abs(got - want).max()))
PY
    python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
