"""A/B of two builds of the library on the same workloads (AIDAX_LIB selects the build per child process).
usage: python scratch/ab_libs.py <other_lib.so> kind:hidden:streams[:form] ..."""
import json, os, subprocess, sys
CHILD = r'''
import importlib, os, sys, tempfile, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
kind, H, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
j = modelgen.make_model(kind, H, 1, seed=H); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
import time
t0 = time.time()
while time.time() - t0 < 0.3:
    for _ in range(16): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 400
e0.record()
for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
e1.record(); torch.cuda.synchronize()
print(pool.kernel_name, round(e0.elapsed_time(e1) / N * 1e3, 2))
'''
other = os.path.abspath(sys.argv[1])
for spec in sys.argv[2:]:
    parts = spec.split(":")
    kind, H, S = parts[0], parts[1], parts[2]
    form = parts[3] if len(parts) > 3 else ""
    row = []
    for label, lib in (("this", ""), ("other", other)):
        env = dict(os.environ)
        if lib: env["AIDAX_LIB"] = lib
        if form: env["AIDAX_KERNEL"] = form
        r = subprocess.run([sys.executable, "-c", CHILD, kind, H, S], env=env, capture_output=True, text=True)
        row.append(f"{label}: {r.stdout.strip() or r.stderr.strip()[-200:]}")
    print(f"{kind}{H} S={S} {form or 'auto'}:  " + "  |  ".join(row), flush=True)
