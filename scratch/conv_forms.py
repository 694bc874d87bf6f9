# conv stack (cfg4 model): fused launch vs packed k_chain launches around the kernel, over stream counts
import importlib, os, sys, tempfile
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
d = tempfile.mkdtemp()
p = W.write_model(W.make_model(kind="conv", hidden=16, input_size=1, seed=1608), os.path.join(d, "c.json"))
for S in (64, 256, 1024, 1536, 2048, 4096, 8192):
    res = []
    for f in ("1", "0", None):
        if f is None: os.environ.pop("AIDAX_CONV_FUSED", None)
        else: os.environ["AIDAX_CONV_FUSED"] = f
        pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
        x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(200): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        res.append((pool.kernel_name, e0.elapsed_time(e1) / 200 * 1e3))
        pool.close()
    print(f"S={S:6d}  fused {res[0][1]:8.1f} us   split {res[1][1]:8.1f} us   default -> {res[2][0]} {res[2][1]:8.1f} us", flush=True)
