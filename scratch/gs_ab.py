# k_gru_gs (recurrent product as bf16 x 3 term products) against k_gru_gm (fp32 MFMAs): cfg3 and GRU-40, pre-rolled clocks
import importlib, os, sys, tempfile, time
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
ctl = dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)
def run(tag, env, hidden, S, n=256, steps=300):
    for k in ("AIDAX_GRU_GM", "AIDAX_GS_PRODUCTS"): os.environ.pop(k, None)
    os.environ.update(env)
    p = W.write_model(W.make_model("gru", hidden, 3, seed=64), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(16): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print("GRU-%d S=%d %-22s %-10s %.1f us" % (hidden, S, tag, pool.kernel_name, e0.elapsed_time(e1) / steps * 1e3), flush=True)
    pool.close()
if os.environ.get("GS_TRIVIAL"):      # the chain passes made trivial: what do the helper waves cost the main waves?
    ctl = dict(eq_bypass=1.0, dc_blocker=0.0, in_lpf_pc=0.0, param1=0.5, param2=0.3)
if os.environ.get("GS_ONLY"):
    run("bf16x3, 6 products", {}, 64, 4096)
    run("bf16x3, 6 products", {}, 40, 4096)
    sys.exit(0)
for hidden, S in ((64, 4096), (64, 8192), (40, 4096)):
    run("fp32 MFMA", {"AIDAX_GRU_GM": "f32"}, hidden, S)
    run("bf16x3, 6 products", {}, hidden, S)
    run("bf16x3, 9 products", {"AIDAX_GS_PRODUCTS": "9"}, hidden, S)
