# cfg3 (GRU-64, 2 params, EQ post) at 4096 streams in the forms that can carry it
import importlib, os, sys, tempfile
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
p = W.write_model(W.make_model("gru", 64, 3, seed=64), os.path.join(tempfile.mkdtemp(), "m.json"))
ctl = dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)
def run(form, S=4096, n=256, steps=100):
    if form: os.environ["AIDAX_KERNEL"] = form
    else: os.environ.pop("AIDAX_KERNEL", None)
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(100): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print(form or "default", pool.kernel_name, "S=%d %.1f us" % (S, e0.elapsed_time(e1) / steps * 1e3), flush=True)
    pool.close()
for S in (2048, 4096, 8192):
    for f in (None, "quad", "mfma", "wave"):
        run(f, S)
