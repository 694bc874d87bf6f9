"""One LV2-sized pool (one stream, 256 frames, the bundled LSTM-12): wall time of 20 000 aidax_pool_process calls as a histogram —
where the occasional slow call of bench.py's realtime_case comes from. Run it plainly for the host-side histogram and once more
under `rocprofv3 --kernel-trace --stats` for the kernel's own duration statistics (profiles/r04_rt_latency.txt)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
path = os.path.join(ROOT, "tests", "golden", "models", "tw40_california_clean_deerinkstudios.json")
pool = ax.Pool(1, 8192, 48000.0)
pool.set_model(ax.Model(path))
x = W.signal(1, 256, seed=5)
for _ in range(500): pool.process(x)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t = np.empty(N)
for i in range(N):
    t0 = time.perf_counter(); pool.process(x); t[i] = time.perf_counter() - t0
t *= 1e6
print(f"{pool.kernel_name}: {N} calls, p50 {np.percentile(t,50):.1f}  p90 {np.percentile(t,90):.1f}  p99 {np.percentile(t,99):.1f}  p99.9 {np.percentile(t,99.9):.1f}  max {t.max():.1f} us")
edges = [0, 50, 52, 54, 56, 58, 60, 65, 70, 80, 100, 150, 1e9]
h, _ = np.histogram(t, edges)
for a, b, c in zip(edges[:-1], edges[1:], h):
    print(f"  {a:5.0f} .. {b if b < 1e8 else float('inf'):5.0f} us: {c:6d}")
slow = np.nonzero(t > 1.5 * np.percentile(t, 50))[0]
print("calls over 1.5 x p50:", len(slow), "; gaps between them (calls):", np.diff(slow)[:20])
pool.close()
