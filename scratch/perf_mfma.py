import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
def run(label, mkw, S, ctl_kw={}, n=256, steps=30):
    j = modelgen.make_model(**mkw)
    p = modelgen.write_model(j, os.path.join(d, label.replace(" ","_")+".json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl_kw))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(3): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label:30s} {pool.kernel_name:16s} S={S:6d}: {ms*1e3:9.1f} us/step  {S*n/ms/1e3:10.1f} Msamples/s", flush=True)
    pool.close()
l96 = dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2)
for S in (2048, 4096, 16384):
    run(f"cfg5 lstm96x2", l96, S, steps=10 if S > 4096 else 30)
run("gru128 1 layer", dict(kind="gru", hidden=128, input_size=3, seed=1), 4096)
run("lstm128 1 layer", dict(kind="lstm", hidden=128, input_size=1, seed=1), 4096)
run("lstm64x2", dict(kind="lstm", hidden=64, input_size=1, seed=1, n_rnn=2), 4096)
os.environ["AIDAX_KERNEL"] = "valu"
run("cfg5 lstm96x2 (valu)", l96, 2048, steps=10)
os.environ.pop("AIDAX_KERNEL")
c16 = dict(kind="conv", hidden=16, input_size=1, seed=1608)
run("cfg4 conv16x8", c16, 1024)
run("cfg4 conv16x8", c16, 8192)
os.environ["AIDAX_KERNEL"] = "valu"
run("cfg4 conv16x8 (valu)", c16, 1024)
