# stacked pools LARGER than one resident grid of k_mfma_ls: one launch per range of streams (aidax_pool.cpp, lp_round_streams)
# against k_mfma (AIDAX_MFMA_LP=0: what such a pool ran on before), pre-rolled clocks
import importlib, os, sys, tempfile, time
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
def run(tag, env, kind, hidden, n_rnn, S, n=256, steps=30):
    for k in ("AIDAX_MFMA_LP",): os.environ.pop(k, None)
    os.environ.update(env)
    p = W.write_model(W.make_model(kind, hidden, 1, seed=hidden, n_rnn=n_rnn), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(4): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    print("%s-%d x%d S=%-6d %-10s %-18s %9.1f us  %.3e samples/s" % (kind, hidden, n_rnn, S, tag, pool.kernel_name, us, S * n / us * 1e6), flush=True)
    pool.close()
MORE = (("lstm", 48, 2), ("gru", 64, 2), ("gru", 96, 2), ("gru", 80, 2), ("lstm", 64, 3), ("gru", 64, 3), ("lstm", 48, 3))
for kind, hidden, n_rnn in MORE if os.environ.get("MORE") else (("lstm", 96, 2), ("lstm", 64, 2), ("gru", 48, 3), ("lstm", 32, 2), ("lstm", 80, 2)):
    for S in (4096, 16384) if os.environ.get("MORE") else (2048, 4096, 8192, 16384):
        run("k_mfma", {"AIDAX_MFMA_LP": "0"}, kind, hidden, n_rnn, S)
        run("default", {}, kind, hidden, n_rnn, S)
