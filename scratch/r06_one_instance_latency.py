"""one-instance pool (the LV2 shell's): aidax_pool_process round trip, back to back, p50 / p99 per block length.
usage: [AIDAX_KERNEL_WORD=0] python scratch/r06_one_instance_latency.py [kind hidden inputs]"""
import ctypes as C, importlib, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ax = importlib.import_module("aidadsp-lv2_amd"); W = ax.workloads
kind, H, I = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("lstm", 16, 1)
path = W.write_model(W.make_model(kind, H, I, seed=H), os.path.join(tempfile.mkdtemp(), "m.json"))
L = ax.lib(); fp = C.POINTER(C.c_float)
out = []
for n in tuple(int(v) for v in os.environ.get("FRAMES", "64,128,256").split(",")):
    pool = ax.Pool(1, n); pool.set_model(ax.Model(path)); pool.set_controls(ax.default_controls(param1=0.4))
    x = (np.random.rand(1, n).astype(np.float32) - 0.5); y = np.empty_like(x)
    px, py = x.ctypes.data_as(fp), y.ctypes.data_as(fp)
    for _ in range(300): L.aidax_pool_process(pool.h, px, py, n)
    t = np.empty(3000)
    for k in range(3000):
        t0 = time.perf_counter(); L.aidax_pool_process(pool.h, px, py, n); t[k] = time.perf_counter() - t0
    out.append(f"{n}: p50 {np.percentile(t, 50) * 1e6:.1f} p99 {np.percentile(t, 99) * 1e6:.1f}")
    name = pool.kernel_name; pool.close()
print(os.environ.get("AIDAX_KERNEL_WORD", "-"), kind, H, I, name, " | ".join(out), flush=True)
