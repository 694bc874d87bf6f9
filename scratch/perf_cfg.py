import importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
def run(cell, H, I, S, ctl_kw, n=256, steps=300, label=""):
    j = modelgen.make_model(cell, H, I, seed=H)
    p = modelgen.write_model(j, os.path.join(d, f"{cell}{H}_{I}.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl_kw))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label:28s} {cell}{H}/{I} S={S:6d} n={n}: {ms*1e3:9.1f} us/step  {S*n/ms/1e3:10.1f} Msamples/s  ({ms*1e6/n:.0f} ns/timestep)", flush=True)
    pool.close()
for S in (256, 1024, 2048, 4096, 16384):
    run("lstm", 32, 1, S, {}, label="default chain")
run("lstm", 32, 1, 1024, dict(net_bypass=1.0), label="chain only (net bypass)")
run("lstm", 32, 1, 1024, dict(in_lpf_pc=0.0, dc_blocker=0.0, eq_bypass=1.0), label="NN + gain ramps only")
run("lstm", 32, 1, 1024, dict(eq_bypass=1.0), label="eq bypassed")
run("lstm", 32, 1, 1024, dict(enabled=0.0), label="disabled (copy)")
run("lstm", 12, 1, 1024, {}, label="lstm12 default")
run("lstm", 64, 1, 1024, {}, label="lstm64 default")
run("gru", 64, 3, 4096, dict(bass_boost_db=4.0), label="cfg3-like gru64/3")
run("gru", 64, 3, 1024, dict(bass_boost_db=4.0), label="gru64/3 1024")
