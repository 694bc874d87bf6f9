"""what a launch of the cfg2 pool costs beyond its frames: kernel time (HIP events over back-to-back launches) against the block length.
usage: python scratch/r06_fixed_cost.py [kind hidden streams]"""
import importlib, os, sys, tempfile, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ax = importlib.import_module("aidadsp-lv2_amd"); W = ax.workloads
kind, H, S = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("lstm", 32, 1024)
p = W.write_model(W.make_model(kind, H, 1, seed=H), os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
res = []
for n in (16, 32, 48, 64, 128, 256):
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        for _ in range(16): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 1000
    e0.record(st)
    for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    res.append((n, e0.elapsed_time(e1) / N * 1e3))
print(pool.kernel_name, S, "streams:", "  ".join(f"{n}: {t:.2f}" for n, t in res))
(n0, t0), (n1, t1) = res[3], res[5]
slope = (t1 - t0) / (n1 - n0)
print(f"per frame {slope * 1e3:.1f} ns; fixed per launch {t1 - slope * n1:.2f} us (from 64 and 256 frames), {res[0][1] - slope * 16:.2f} us (at 16 frames)")
