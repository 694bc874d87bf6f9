#!/bin/bash
# tanh(c) of the LSTM cell in the exp form (5 instructions) against the rational (15), round 5: cfg2 time, the goldens' self-test, a 48 128-sample drift
cd "$(dirname "$0")/.."
for v in "$@"; do
  export AIDAX_LIB=$PWD/${v#*=}
  echo "=== ${v%%=*}"
  python bench.py --workload cfg2 --no-others --no-cpu-baseline --no-traffic --no-dist --steps 3000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['roofline']['kernel_ms']*1e3,2), 'us', d['max_abs_err'])"
  python - <<'PY'
import importlib, os, sys, glob
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
for f in sorted(glob.glob("tests/golden/models/*.json")):
    m = ax.Model(f); r = m.self_test(); print("  ", os.path.basename(f)[:34], "self-test", r)
os.makedirs("build", exist_ok=True)
for name, path in (("lstm32", modelgen.write_model(modelgen.make_model("lstm", 32, 1, seed=32), "build/l32.json")), ("british_lead", [g for g in glob.glob("tests/golden/models/*british*")][0])):
    spec = O.load_model(path); pool = ax.Pool(2, 256); pool.set_model(ax.Model(path)); pool.set_controls(ax.default_controls())
    x = modelgen.signal(2, 48128, seed=5); got = np.concatenate([pool.process(np.ascontiguousarray(x[:, b:b+256])) for b in range(0, 48128, 256)], axis=1)
    want = O.run_streams(spec, O.default_controls(), x, 256)
    print("  ", name, pool.kernel_name, "48128-sample drift max err", float(np.abs(got - want).max()))
PY
done
