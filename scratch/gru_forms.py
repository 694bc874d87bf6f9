"""GRU models of the reference's table over stream counts, this build against another (AIDAX_LIB): what the exp-form candidate bought.
usage: python scratch/gru_forms.py [other_lib.so]"""
import json, os, subprocess, sys
CHILD = r'''
import importlib, os, sys, tempfile, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
res = {}
for H, I in ((8, 1), (16, 1), (24, 2), (40, 1), (64, 3), (80, 1)):
    j = modelgen.make_model("gru", H, I, seed=H); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
    for S in (1, 1024, 4096):
        pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
        x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        for _ in range(150): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 600 if S < 4096 else 300
        e0.record()
        for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        e1.record(); torch.cuda.synchronize()
        res[f"gru{H}/{I} x{S}"] = (pool.kernel_name, round(e0.elapsed_time(e1) / N * 1e3, 1))
        pool.close()
import json; print(json.dumps(res))
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
