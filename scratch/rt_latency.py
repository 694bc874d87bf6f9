#!/usr/bin/env python3
"""Wall time of the audio-thread entry points of the LV2 shell (one instance, 256-frame blocks, the bundled
LSTM-12 model and a synthetic LSTM-32): run() p50 / p99 / max, work() (worker) and work_response() (audio).
Run once per setting of AIDAX_ZEROCOPY:  python scratch/rt_latency.py   |   AIDAX_ZEROCOPY=0 python scratch/rt_latency.py
Numbers quoted in INTEGRATION.md."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AIDAX_NO_TORCH"] = "1"
import numpy as np  # noqa: E402

from tests import lv2host, modelgen  # noqa: E402

bundle = tempfile.mkdtemp(prefix="aidax_rt_")
src = os.path.join(ROOT, "tests", "golden", "models")
dst = os.path.join(bundle, "models", "deer ink studios")
os.makedirs(dst)
for f in os.listdir(src):
    shutil.copy(os.path.join(src, f), dst)
modelgen.write_model(modelgen.make_model("lstm", 32, 1, seed=32), os.path.join(bundle, "models", "lstm32.json"))

out = {"zero_copy": os.environ.get("AIDAX_ZEROCOPY", "1") != "0", "cases": []}
for label, rel in (("LSTM-12 bundled", "models/deer ink studios/tw40_california_clean_deerinkstudios.json"),
                   ("LSTM-32 synthetic", "models/lstm32.json")):
    h = lv2host.Host(bundle_dir=bundle)
    x = modelgen.signal(1, 256, seed=1)[0]
    h.run(x)
    h.restore(rel)
    t0 = time.perf_counter(); h.pump_worker(); t_work = time.perf_counter() - t0
    h.run(x)
    t0 = time.perf_counter(); h.deliver_responses(); t_resp = time.perf_counter() - t0
    h.pump_worker()
    for _ in range(500):
        h.run(x)
    t = np.empty(5000)
    for i in range(t.size):
        h.audio_in[:256] = x
        t0 = time.perf_counter()
        h.desc.run(h.handle, 256)
        t[i] = time.perf_counter() - t0
    out["cases"].append({"model": label, "run_us": {"p50": float(np.percentile(t, 50) * 1e6), "p99": float(np.percentile(t, 99) * 1e6),
                                                     "p999": float(np.percentile(t, 99.9) * 1e6), "max": float(t.max() * 1e6)},
                         "work_ms": t_work * 1e3, "work_response_us": t_resp * 1e6,
                         "period_us": 256 / 48000.0 * 1e6})
    h.close()
print(json.dumps(out))
