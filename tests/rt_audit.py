#!/usr/bin/env python3
"""Audit driver for tests/test_rt_contract.py. Runs in a process of its own with tests/hip_audit.c preloaded
(LD_PRELOAD) and counts, per phase, the HIP runtime calls made on the calling thread:

    * the LV2 plugin's life cycle through the mock host: run() before a model, work() (worker thread),
      work_response() (audio thread), run() with a model, a second model through patch:Set, activate();
    * the C ABI underneath: aidax_pool_prepare_model / aidax_pool_commit_model / aidax_pool_process_device;
    * hub mode: aidax_hub_run of two instances.

Prints one json object {phase: {hip call: count}} on the last line of stdout. No torch in this process
(AIDAX_NO_TORCH=1): exactly one HIP runtime, the one the shim forwards to."""
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AIDAX_NO_TORCH"] = "1"

import numpy as np  # noqa: E402

from tests import lv2host, modelgen  # noqa: E402

shim = C.CDLL(os.environ["HIP_AUDIT_LIB"])
shim.hip_audit_name.restype = C.c_char_p
NF = shim.hip_audit_fields()
NAMES = [shim.hip_audit_name(i).decode() for i in range(NF)]
report = {}


class audit:
    def __init__(self, phase):
        self.phase = phase

    def __enter__(self):
        shim.hip_audit_begin()

    def __exit__(self, *a):
        out = (C.c_uint64 * NF)()
        shim.hip_audit_end(out)
        acc = report.setdefault(self.phase, {n: 0 for n in NAMES})
        acc["calls"] = acc.get("calls", 0) + 1
        for n, v in zip(NAMES, out):
            acc[n] += int(v)


def lv2_life_cycle(bundle):
    h = lv2host.Host(bundle_dir=bundle)
    assert h.handle
    x = modelgen.signal(1, 256 * 12, seed=3)[0]
    blk = lambda i: x[i * 256:(i + 1) * 256]
    with audit("run_no_model"):
        h.run(blk(0))
    h.restore("models/deer ink studios/tw40_california_clean_deerinkstudios.json")
    with audit("work_load"):
        assert h.pump_worker() == 1
    with audit("work_response"):
        assert h.deliver_responses() == 1
    with audit("work_free"):
        h.pump_worker()
    for i in range(1, 5):
        with audit("run_model"):
            y = h.run(blk(i))
    assert np.abs(y).max() > 1e-4
    h.controls(BASS=3.0, PREGAIN=2.0)
    with audit("run_controls_changed"):
        h.run(blk(5))
    h.send_patch_set(os.path.join(bundle, "models", "gru16.json"))
    with audit("run_patch_set"):
        h.run(blk(6))
    with audit("work_load"):
        assert h.pump_worker() == 1
    with audit("work_response"):
        assert h.deliver_responses() == 1
    with audit("work_free"):
        h.pump_worker()
    for i in range(7, 10):
        with audit("run_model"):
            h.run(blk(i))
    with audit("activate"):
        h.desc.activate(h.handle)
    with audit("run_model"):
        h.run(blk(10))
    with audit("run_pre_run"):
        h.run(np.zeros(0, np.float32))
    h.close()


def c_abi(bundle):
    ax = importlib.import_module("aidadsp-lv2_amd")
    m1 = ax.Model(os.path.join(bundle, "models", "gru16.json"))
    m2 = ax.Model(os.path.join(bundle, "models", "lstm32.json"))
    pool = ax.Pool(512, 256)
    pool.set_model(m1)
    x = modelgen.signal(512, 256, seed=4)
    pool.process(x)
    with audit("abi_prepare"):
        sg = pool.prepare_model(m2)
    with audit("abi_commit"):
        pool.commit_model(sg)
    with audit("abi_set_controls"):
        pool.set_controls(ax.default_controls(bass_boost_db=2.0))
    with audit("abi_process"):
        pool.process(x)
    with audit("abi_staged_free"):
        pool.staged_free(sg)
    pool.close()

    hub = ax.Hub(8, 256)
    hub.set_model(m1)
    hub.set_deadline_us(0)
    with audit("hub_attach"):
        a, b = hub.attach(), hub.attach()
    for k in range(6):
        with audit("hub_run"):
            hub.run(a, x[0])
            hub.run(b, x[1])
        time.sleep(0.003)          # the rest of the host's audio period: the launcher thread closes the period meanwhile
    hub.close()


if __name__ == "__main__":
    bundle = sys.argv[1]
    lv2_life_cycle(bundle)
    c_abi(bundle)
    print(json.dumps(report))
