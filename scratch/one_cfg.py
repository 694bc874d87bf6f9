"""One bench workload, N passes (for rocprofv3 --kernel-trace --stats): python scratch/one_cfg.py cfg5 100"""
import importlib, os, sys, tempfile, time, torch
sys.path.insert(0, os.getcwd())
import bench
ax = importlib.import_module("aidadsp-lv2_amd")
from tests import modelgen
name, N = sys.argv[1], int(sys.argv[2])
w = bench.WORKLOADS[name]
j = modelgen.make_model(**w["model"]); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
S = w["streams"]
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
if w.get("controls"): pool.set_controls(ax.default_controls(**w["controls"]))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
t0 = time.time()
while time.time() - t0 < 0.4:
    for _ in range(8): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
    torch.cuda.synchronize()
for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
