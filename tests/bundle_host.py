#!/usr/bin/env python3
"""Loads the plugin the way an LV2 host would — by the binary named in the bundle's manifest.ttl, from the bundle
directory, with the default state of the bundle's rt-neural-generic.ttl restored relative to it — in a process
that has never seen the in-tree build, and compares the audio with the CPU oracle (tests/test_bundle.py).

    bundle_host.py <bundle dir>      prints one json line: max_abs_err, the shared objects that were mapped, ...
"""
import json
import os
import re
import sys
import urllib.parse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AIDAX_NO_TORCH"] = "1"

import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tests import lv2host, modelgen  # noqa: E402

bundle = os.path.abspath(sys.argv[1])
manifest = open(os.path.join(bundle, "manifest.ttl")).read()
binary = re.search(r"lv2:binary\s+<([^>]+)>", manifest).group(1)
ttl_name = re.search(r"rdfs:seeAlso\s+<([^>]+)>", manifest).group(1)
ttl = open(os.path.join(bundle, ttl_name)).read()
state_rel = urllib.parse.unquote(re.search(r"state:state\s*\[\s*<[^>]+>\s*<([^>]+)>", ttl).group(1))

h = lv2host.Host(bundle_dir=bundle, so=os.path.join(bundle, binary))
assert h.handle, "instantiate failed"
plug = O.OraclePlugin()
x = modelgen.signal(1, 256 * 8, seed=19)[0]
err = 0.0
assert h.restore(state_rel) == 0                      # what a host does with state:loadDefaultState
assert h.pump_worker() == 1
h.run(np.zeros(0, np.float32)); plug.run(O.default_controls(), np.zeros(0, np.float32))
assert h.deliver_responses() == 1
h.pump_worker()
plug.set_model(O.OracleModel(O.load_model(os.path.join(bundle, state_rel)), 0.0, 0.0))
latency = lv2host.C.c_float(-1.0)
h.desc.connect_port(h.handle, 25, lv2host.C.cast(lv2host.C.pointer(latency), lv2host.C.c_void_p))
peak = 0.0
for b in range(8):
    blk = x[b * 256:(b + 1) * 256]
    got = h.run(blk)
    want = plug.run(O.default_controls(), blk)
    err = max(err, float(np.abs(got - want).max()))
    peak = max(peak, float(np.abs(got).max()))
maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "aidax" in ln or "rt-neural-generic" in ln})
rc, stored = h.save()
h.close()
print(json.dumps({"max_abs_err": err, "peak": peak, "maps": maps, "state": state_rel, "latency": latency.value,
                  "model_in_size": h.ctl["ModelInSize"].value,
                  "saved": [s[1].rstrip(b"\0").decode() for s in stored]}))
