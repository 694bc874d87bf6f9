"""How much of a pipeline pass is the helper waves'? The same model and stream count with the chain passes made trivial
(EQ bypassed: the post cascade is one stage instead of six; LPF / DC blocker out of circuit)."""
import importlib, os, sys, tempfile, time, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
def run(kind, H, S, **ckw):
    j = modelgen.make_model(kind, H, 1, seed=H); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ckw))
    x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(16): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 600
    e0.record()
    for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print(f"{kind}{H} S={S} {ckw or 'TTL defaults'}: {pool.kernel_name} {e0.elapsed_time(e1) / N * 1e3:.2f} us", flush=True)
    pool.close()
for H in (32, 12):
    for S in (1024, 1):
        run("lstm", H, S)
        run("lstm", H, S, eq_bypass=1.0)
        run("lstm", H, S, eq_bypass=1.0, dc_blocker=0.0, in_lpf_pc=0.0)
