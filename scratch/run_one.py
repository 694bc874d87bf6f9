# usage: run_one.py <kind> <hidden> <streams> [steps]   (AIDAX_KERNEL picks the form) - a fixed workload for rocprofv3
import importlib, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
kind, H, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
p = modelgen.write_model(modelgen.make_model(kind, H, 1, seed=H), os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
torch.cuda.synchronize()
print(pool.kernel_name)
