import importlib, os, sys, tempfile
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
p = W.write_model(W.make_model("conv", 16, 1, seed=1608), os.path.join(tempfile.mkdtemp(), "m.json"))
def run(S=1024, n=256, steps=300, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(net_bypass=0.0))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(300): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print(env, pool.kernel_name, "S=%d %.1f us" % (S, e0.elapsed_time(e1) / steps * 1e3), flush=True)
    pool.close()
for t in sys.argv[1:] or ["0"]:
    run(env={"AIDAX_TUNE": t})
