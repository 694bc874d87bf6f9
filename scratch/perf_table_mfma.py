import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
def run(label, mkw, S, ctl_kw={}, n=256, steps=20):
    j = modelgen.make_model(**mkw)
    p = modelgen.write_model(j, os.path.join(d, label.replace(" ","_")+".json"))
    res = []
    for form in ("", os.environ.get("ALT", "mfma")):
        if form: os.environ["AIDAX_KERNEL"] = form
        else: os.environ.pop("AIDAX_KERNEL", None)
        pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl_kw))
        x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        for _ in range(3): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        e1.record(); torch.cuda.synchronize()
        res.append((pool.kernel_name, e0.elapsed_time(e1) / steps * 1e3))
        pool.close()
    print(f"{label:16s} S={S:6d}: {res[0][0]:22s} {res[0][1]:8.1f} us | {res[1][0]:16s} {res[1][1]:8.1f} us   ratio {res[0][1]/res[1][1]:.2f}", flush=True)
import itertools
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 4096, 8192, 16384]
hs = [int(a) for a in os.environ.get("HS", "8,16,32,64,80").split(",")]
for kind, H in itertools.product(("lstm", "gru"), hs):
    for S in sizes:
        run(f"{kind}{H}", dict(kind=kind, hidden=H, input_size=1, seed=H), S, steps=10)
