"""A/B of two builds of the library on the bench's other workloads (cfg3 / cfg4 / cfg5), clocks pre-rolled like bench.py.
usage: python scratch/ab_cfg.py <other_lib.so> cfg3 cfg5 ...   (AIDAX_* env passes through to both)"""
import json, os, subprocess, sys
CHILD = r'''
import importlib, os, sys, tempfile, time, torch
sys.path.insert(0, os.getcwd())
import bench
ax = importlib.import_module("aidadsp-lv2_amd")
from tests import modelgen
name = sys.argv[1]
w = bench.WORKLOADS[name]
j = modelgen.make_model(**w["model"]); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
S = w["streams"]
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
if w.get("controls"): pool.set_controls(ax.default_controls(**w["controls"]))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
t0 = time.time()
while time.time() - t0 < 0.5:
    for _ in range(8): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
    torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = int(sys.argv[2])
    e0.record()
    for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    res.append(round(e0.elapsed_time(e1) / N * 1e3, 1))
print(pool.kernel_name, res)
'''
other = os.path.abspath(sys.argv[1])
for name in sys.argv[2:]:
    row = []
    for label, lib in (("this", ""), ("other", other), ("this", ""), ("other", other)):
        env = dict(os.environ)
        if lib: env["AIDAX_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", CHILD, name, {"cfg5": "150", "cfg3": "400"}.get(name, "1500")], env=env, capture_output=True, text=True)
        row.append(f"{label}: {r.stdout.strip() or r.stderr.strip()[-300:]}")
    print(f"{name}:  " + "  |  ".join(row), flush=True)
