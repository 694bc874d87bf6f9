# one process, several AIDAX_TUNE / env settings of the cfg5 pass (fresh box python start-up is paid once)
import importlib, os, sys, tempfile
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
p = W.write_model(W.make_model("lstm", 96, 1, seed=96, n_rnn=2), os.path.join(tempfile.mkdtemp(), "m.json"))
def run(S=2048, n=256, steps=100, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(env, pool.kernel_name, "S=%d %.1f us  %.1f%% of MFMA peak" % (S, ms * 1e3, 222144 * S * n / (ms * 1e-3) / 157.3e12 * 100), flush=True)
    pool.close()
for t in sys.argv[1:] or ["0"]:
    run(env={"AIDAX_TUNE": t})
