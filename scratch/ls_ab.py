# k_mfma_ls (contractions as bf16 term products of split operands) against k_mfma_lp (fp32 MFMAs): stacked models, pre-rolled clocks
import importlib, os, sys, tempfile, time
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
def run(tag, env, kind, hidden, n_rnn, S, n=256, steps=100):
    for k in ("AIDAX_LP_SPLIT",): os.environ.pop(k, None)
    os.environ.update(env)
    p = W.write_model(W.make_model(kind, hidden, 1, seed=hidden, n_rnn=n_rnn), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(8): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print("%s-%d x%d S=%d %-20s %-18s %.1f us" % (kind, hidden, n_rnn, S, tag, pool.kernel_name, e0.elapsed_time(e1) / steps * 1e3), flush=True)
    pool.close()
cases = [("lstm", 96, 2, 2048)] if os.environ.get("LS_ONLY") else [("lstm", 96, 2, 2048), ("lstm", 64, 2, 2048), ("gru", 48, 3, 1024), ("lstm", 32, 2, 2048), ("lstm", 80, 2, 2048)]
for kind, hidden, n_rnn, S in cases:
    run("fp32 MFMA", {"AIDAX_LP_SPLIT": "0"}, kind, hidden, n_rnn, S)
    run("bf16x3, 6 products", {}, kind, hidden, n_rnn, S)
