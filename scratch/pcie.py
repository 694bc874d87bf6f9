"""PCIe-inclusive rate of a cfg2 block through the host-buffer entry points (pageable numpy buffers):
aidax_pool_process (copy into pinned staging + H2D + pass + D2H + wait + copy out, one after the other) against
aidax_pool_submit / aidax_pool_collect with one block kept in flight (the copies of the neighbouring blocks run under
the pass). DESIGN.md quotes both; neither is ever bench.py's `value`."""
import importlib, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
j = modelgen.make_model("lstm", 32, 1, seed=32); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(1024, 256); pool.set_model(ax.Model(p))
xs = [modelgen.signal(1024, 256, seed=s) for s in range(4)]
for _ in range(20): pool.process(xs[0])
N = 400
t0 = time.perf_counter()
for k in range(N): pool.process(xs[k & 3])
dt = (time.perf_counter() - t0) / N
print(f"aidax_pool_process (blocking, pageable numpy): {dt*1e6:.1f} us/block -> {1024*256/dt/1e9:.3f} Gsamples/s")
out = np.empty((1024, 256), np.float32)
pool.submit(xs[0])
for k in range(1, 40): pool.submit(xs[k & 3]); pool.collect(256, out)
t0 = time.perf_counter()
for k in range(N): pool.submit(xs[k & 3]); pool.collect(256, out)
dt = (time.perf_counter() - t0) / N
pool.collect(256, out)
print(f"aidax_pool_submit / collect (one block in flight):  {dt*1e6:.1f} us/block -> {1024*256/dt/1e9:.3f} Gsamples/s")
