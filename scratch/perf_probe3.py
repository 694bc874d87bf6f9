import importlib, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import torch
B = importlib.import_module("aidadsp-lv2_amd.binding")
if len(sys.argv) > 1 and sys.argv[1] != "0":
    path = os.path.join(os.getcwd(), "scratch", "probe", sys.argv[1], "libaidax_hip.so")
    B.lib_path = lambda: path
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
def run(label, mkw, S, n=256, steps=300):
    p = modelgen.write_model(modelgen.make_model(**mkw), os.path.join(d, "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print(f"variant {sys.argv[1] if len(sys.argv)>1 else 0}: {label:12s} {pool.kernel_name:18s} S={S:6d}: {e0.elapsed_time(e1)/steps*1e3:9.2f} us/step", flush=True)
    pool.close()
run("lstm32", dict(kind="lstm", hidden=32, input_size=1, seed=32), 1024)
run("lstm12", dict(kind="lstm", hidden=12, input_size=1, seed=12), 1024)
run("gru24/3", dict(kind="gru", hidden=24, input_size=3, seed=24), 1024)
