"""One-layer models on k_mfma_lp (a workgroup per 16 streams, fragments resident, no ring) against what the pool picks
today, k_mfma and k_quad: the sweep behind many_streams_form (aidax_pool.cpp). Usage: perf_lp1.py [streams,...]  HS=... KINDS=..."""
import importlib, os, sys, tempfile, itertools
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
FORMS = [("auto", {}), ("mfma_lp", {"AIDAX_KERNEL": "mfma", "AIDAX_MFMA_LP": "1"}), ("mfma", {"AIDAX_KERNEL": "mfma", "AIDAX_MFMA_LP": "0"}),
         ("quad", {"AIDAX_KERNEL": "quad"})]
def run(label, mkw, S, ctl_kw={}, n=256, steps=12):
    j = modelgen.make_model(**mkw)
    p = modelgen.write_model(j, os.path.join(d, label.replace(" ", "_") + ".json"))
    res = []
    for name, env in FORMS:
        for k in ("AIDAX_KERNEL", "AIDAX_MFMA_LP"): os.environ.pop(k, None)
        os.environ.update(env)
        try:
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
        except Exception as e:
            res.append(("-", float("nan")))
    best = min(range(len(res)), key=lambda i: res[i][1] if res[i][1] == res[i][1] else 1e30)
    print(f"{label:10s} S={S:6d}: " + " | ".join(f"{FORMS[i][0]}={res[i][0].replace('k_chain+','')}:{res[i][1]:7.1f}" for i in range(len(res))) + f"   best {FORMS[best][0]}", flush=True)
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 1024, 2048, 4096]
hs = [int(a) for a in os.environ.get("HS", "16,32,40,64,80").split(",")]
kinds = os.environ.get("KINDS", "lstm,gru").split(",")
if os.environ.get("CFG3"):
    run("cfg3", dict(kind="gru", hidden=64, input_size=3, seed=64), 4096,
        dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3), steps=30)
for kind, H in itertools.product(kinds, hs):
    for S in sizes:
        run(f"{kind}{H}", dict(kind=kind, hidden=H, input_size=1, seed=H), S)
