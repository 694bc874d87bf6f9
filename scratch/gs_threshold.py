# where does k_gru_gs start to win? GRU-64 / GRU-40 (2 params, EQ) over stream counts: the pool's choice against AIDAX_KERNEL=mfma (k_gru_gs) and =quad
import importlib, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
ctl = dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)
def run(hidden, form, S, n=256, steps=200):
    if form: os.environ["AIDAX_KERNEL"] = form
    else: os.environ.pop("AIDAX_KERNEL", None)
    p = W.write_model(W.make_model("gru", hidden, 3, seed=64), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    t0 = time.time()
    while time.time() - t0 < 0.2:
        for _ in range(16): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    name = pool.kernel_name
    pool.close()
    return name, us
for hidden in (64, 40):
    for S in (256, 512, 1024, 1536, 2048, 2560, 3072, 3584, 4096):
        row = [f"GRU-{hidden} S={S:5d}"]
        for form in (None, "mfma", "quad"):
            name, us = run(hidden, form, S)
            row.append(f"{form or 'default':7s} {name:18s} {us:7.1f} us")
        print("  |  ".join(row), flush=True)
