# one-layer models of the reference's table: k_mfma_ls1 (AIDAX_KERNEL=mfma AIDAX_LS1=1: bf16 term products) against what the pool
# picks (auto) and against the fp32 matrix-core forms (AIDAX_KERNEL=mfma), 256-frame blocks, pre-rolled clocks
import importlib, os, sys, tempfile, time
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
def run(env, kind, hidden, S, n=256, steps=40):
    for k in ("AIDAX_KERNEL", "AIDAX_LS1", "AIDAX_GRU_GM", "AIDAX_LSTM_GS"): os.environ.pop(k, None)
    os.environ.update(env)
    p = W.write_model(W.make_model(kind, hidden, 1, seed=hidden), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        for _ in range(8): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    name = pool.kernel_name
    pool.close()
    return name, e0.elapsed_time(e1) / steps * 1e3
shapes = [("lstm", h) for h in (16, 32, 40, 64, 80)] + [("gru", h) for h in (16, 32, 40, 64, 80)]
if os.environ.get("LS1_SHAPES"): shapes = [tuple(s.split("-")) for s in os.environ["LS1_SHAPES"].split(",")]; shapes = [(k, int(h)) for k, h in shapes]
for kind, hidden in shapes:
    for S in (256, 1024, 2048, 4096, 8192, 16384):
        cells = []
        for tag, env in (("auto", {}), ("mfma", {"AIDAX_KERNEL": "mfma", "AIDAX_GRU_GM": "0"}), ("ls1", {"AIDAX_KERNEL": "mfma", "AIDAX_LS1": "1", "AIDAX_GRU_GM": "0"})) + ((("lgs", {"AIDAX_KERNEL": "mfma", "AIDAX_LSTM_GS": "1"}),) if kind == "lstm" and hidden in (40, 64, 80) else (("gs", {"AIDAX_KERNEL": "mfma"}),) if kind == "gru" and hidden == 80 else ()):
            name, us = run(env, kind, hidden, S)
            cells.append("%s=%s: %7.1f" % (tag, name, us))
        print("%s%-3d S=%6d: %s" % (kind, hidden, S, " | ".join(cells)), flush=True)
