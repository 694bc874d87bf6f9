import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
def run(cell, H, I, S, n=256, steps=100):
    j = modelgen.make_model(cell, H, I, seed=H)
    p = modelgen.write_model(j, os.path.join(d, f"{cell}{H}_{I}.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(10): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{cell}{H}/{I} {pool.kernel_name:22s} S={S:6d}: {ms*1e3:9.1f} us  {S*n/ms/1e3:10.1f} Msamples/s", flush=True)
    pool.close()
for H in (8, 12, 16, 20, 24, 32, 40, 64, 80):
    run("gru", H, 1, 1024)
for H in (16, 32, 80):
    run("gru", H, 1, 8192, steps=30)
