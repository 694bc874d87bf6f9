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
p = modelgen.write_model(modelgen.make_model("lstm", 32, 1, seed=32), os.path.join(d, "m.json"))
S, n = 1024, 256
pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(200): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5000): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print(f"variant {sys.argv[1] if len(sys.argv)>1 else 0}: cfg2 {e0.elapsed_time(e1)/5000*1e3:8.3f} us/step", flush=True)
