"""cfg2 blocks from and to registered host buffers, two in flight (bench.py host_inclusive's last form) — run under
rocprofv3 --kernel-trace --memory-copy-trace to see where a block's time above the pass goes.
usage: python scratch/r06_host_pipe_trace.py [blocks] [in_flight]"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ax = importlib.import_module("aidadsp-lv2_amd"); W = ax.workloads
n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 300
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
path, _ = bench.workload_model_path(W, "cfg2")
S, N = bench.WORKLOADS["cfg2"]["streams"], bench.N_FRAMES
L = ax.lib(); fp = C.POINTER(C.c_float)
arena = np.zeros((8, S, N), np.float32)
for r in range(4): arena[r] = W.signal(S, N, seed=0xA1DA + r)
pin = [arena[r].ctypes.data_as(fp) for r in range(4)]; pout = [arena[4 + r].ctypes.data_as(fp) for r in range(4)]
pool = ax.Pool(S, N, 48000.0, device=0); pool.set_model(ax.Model(path)); pool.set_controls(ax.default_controls()); h = pool.h
pool.register_host(arena)
def ok(rc):
    if rc < 0: raise SystemExit(L.aidax_last_error().decode())
for k in range(depth): ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N))
for k in range(depth, 40):
    ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N)); ok(L.aidax_pool_collect(h, pout[(k - depth) % 4], N))
t0 = time.perf_counter()
for k in range(40, 40 + n_blocks):
    ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N)); ok(L.aidax_pool_collect(h, pout[(k - depth) % 4], N))
dt = time.perf_counter() - t0
for k in range(40 + n_blocks - depth, 40 + n_blocks): ok(L.aidax_pool_collect(h, pout[k % 4], N))
print(f"depth {depth}: {dt / n_blocks * 1e6:.2f} us per block", flush=True)
pool.unregister_host(arena); pool.close()
