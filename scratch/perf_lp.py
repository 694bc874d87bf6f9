# k_mfma_lp vs k_mfma (AIDAX_MFMA_LP=0) over stream counts and stacked shapes; numbers quoted in DESIGN.md
import importlib, os, sys, tempfile
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
d = tempfile.mkdtemp()
def run(label, mkw, S, n=256, steps=30):
    p = W.write_model(W.make_model(**mkw), os.path.join(d, label.replace(" ", "_") + ".json"))
    res = []
    for lp in ("1", "0"):
        os.environ["AIDAX_MFMA_LP"] = lp
        pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
        x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        for _ in range(5): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        res.append((pool.kernel_name, e0.elapsed_time(e1) / steps))
        pool.close()
    print(f"{label:18s} S={S:6d}: {res[0][0]:18s} {res[0][1]*1e3:9.1f} us   {res[1][0]:16s} {res[1][1]*1e3:9.1f} us   x{res[1][1]/res[0][1]:.2f}", flush=True)
l96 = dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2)
for S in (256, 1024, 2048, 4096, 8192, 16384):
    run("lstm96x2", l96, S, steps=10 if S > 4096 else 30)
for S in (2048, 8192):
    run("lstm64x2", dict(kind="lstm", hidden=64, input_size=1, seed=1, n_rnn=2), S, steps=20)
    run("gru48x3", dict(kind="gru", hidden=48, input_size=2, seed=1, n_rnn=3), S, steps=20)
    run("lstm32x2", dict(kind="lstm", hidden=32, input_size=1, seed=1, n_rnn=2), S, steps=20)
