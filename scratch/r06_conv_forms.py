"""Round 6: the conv stacks with a compiled k_conv_st geometry at a host's block lengths, streaming form against layer-major form
(hooks build: AIDAX_CONV_ST=0 sends every block through k_conv_ms), 1024 streams, kernel time from HIP events on the launch stream.
usage: python scratch/r06_conv_forms.py  ->  profiles/r06_conv_forms.txt"""
import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AIDAX_LIB", os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "hooks", "libaidax_hip.so"))
import torch
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
d = tempfile.mkdtemp()
STACKS = [("A cfg4: 8 x k3, 1..128", dict(seed=1608)),
          ("B 10 x k2, two cycles 1..16", dict(seed=3, conv_k=2, conv_dilations=[1, 2, 4, 8, 16] * 2)),
          ("C 6 x k3, 1..32", dict(seed=3, conv_layers=6)),
          ("D 8 x k3, two cycles 1..8", dict(seed=3, conv_dilations=[1, 2, 4, 8] * 2)),
          ("E 8 x k2, 1..128", dict(seed=3, conv_layers=8, conv_k=2))]
S = int(os.environ.get("STREAMS", "1024"))
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
print(f"# {S} streams, full run() chain, TTL-default controls; us per launch (average of 300 after 40 warm-up launches), hooks build")
print(f"{'stack':34s} {'frames':>6s} {'k_conv_st':>10s} {'k_conv_ms':>10s} {'ratio':>6s}  samples/s (st)")
for name, kw in STACKS:
    p = W.write_model(W.make_model(kind="conv", hidden=16, input_size=1, **kw), os.path.join(d, "c.json"))
    for n in (64, 128, 256):
        res = {}
        for flag in ("1", "0"):
            os.environ["AIDAX_CONV_ST"] = flag
            pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
            x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
            for _ in range(40): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(300): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
            e1.record(st); torch.cuda.synchronize()
            res[flag] = e0.elapsed_time(e1) / 300 * 1e3
            assert pool.kernel_name == ("k_conv_st" if flag == "1" else "k_conv_ms"), pool.kernel_name
            pool.close()
        print(f"{name:34s} {n:6d} {res['1']:10.1f} {res['0']:10.1f} {res['0'] / res['1']:6.2f}  {S * n / res['1'] * 1e6:.3e}", flush=True)
