import importlib, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
j = modelgen.make_model("lstm", 32, 1, seed=32); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(1024, 256); pool.set_model(ax.Model(p))
x = modelgen.signal(1024, 256)
for _ in range(20): pool.process(x)
t0 = time.perf_counter(); N = 300
for _ in range(N): pool.process(x)
dt = (time.perf_counter() - t0) / N
print(f"host-buffer process (H2D + kernel + D2H + sync, pageable numpy): {dt*1e6:.1f} us/block -> {1024*256/dt/1e9:.3f} Gsamples/s")
