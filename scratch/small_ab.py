"""Small cells of the reference's table in the pipeline form, this build against another (AIDAX_LIB). usage: python scratch/small_ab.py other.so"""
import json, os, subprocess, sys
CHILD = r'''
import importlib, os, sys, tempfile, torch, json
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
res = {}
for kind, H, I in (("lstm", 8, 1), ("lstm", 12, 1), ("lstm", 16, 1), ("lstm", 24, 2), ("lstm", 32, 1), ("gru", 8, 1), ("gru", 16, 3)):
    j = modelgen.make_model(kind, H, I, seed=H); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
    for S in (1, 256, 1024, 2048):
        pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
        x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        for _ in range(200): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 800
        e0.record()
        for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        e1.record(); torch.cuda.synchronize()
        res[f"{kind}{H}/{I} x{S}"] = (pool.kernel_name, round(e0.elapsed_time(e1) / N * 1e3, 1))
        pool.close()
print(json.dumps(res))
'''
rows = []
for lib in ([""] + sys.argv[1:2]):
    env = dict(os.environ)
    if lib: env["AIDAX_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    rows.append(json.loads(r.stdout.strip().splitlines()[-1]))
for k in rows[0]:
    line = f"{k:18s} {rows[0][k][0]:24s} {rows[0][k][1]:8.1f} us"
    if len(rows) > 1: line += f"   other: {rows[1][k][0]:24s} {rows[1][k][1]:8.1f} us   ({(rows[0][k][1] / rows[1][k][1] - 1) * 100:+.1f} %)"
    print(line)
