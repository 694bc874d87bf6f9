"""When do the 1024 workgroups of cfg2's pipeline launch start and end? Measurement build: make OBJDIR=build/obj_pt LIBDIR=build/lib_pt
EXTRA=-DAIDAX_PIPE_TRACE build/lib_pt/libaidax_hip.so; AIDAX_LIB=build/lib_pt/libaidax_hip.so python scratch/pipe_trace.py
Wave Q of every workgroup leaves the chip's 100 MHz clock at start and end, its shader cycles and HW ids in its output row."""
import importlib, os, sys, tempfile, collections
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
j = modelgen.make_model("lstm", 32, 1, seed=32); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
S = 1024
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(300): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
raw = y[:, :5].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
np.save("gpurun_out/pipe_trace.npy", raw)
w0, w1, cyc = raw[:, 0], raw[:, 1], raw[:, 2]
hw, xcc = raw[:, 3], raw[:, 4] & 0xf
print(f"first start -> last start {(w0.max() - w0.min()) / 100:.2f} us; first start -> last end {(w1.max() - w0.min()) / 100:.2f} us")
d = (w1 - w0) / 100
print(f"per workgroup start -> end: p10 {np.percentile(d, 10):.2f} median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f} us; shader cycles median {np.median(cyc):.0f}")
end = (w1 - w0.min()) / 100
print("end time by workgroup id quartile (median):", [round(float(np.median(end[q * 256:(q + 1) * 256])), 2) for q in range(4)])
print("duration by workgroup id quartile (median):", [round(float(np.median(d[q * 256:(q + 1) * 256])), 2) for q in range(4)])
cu = xcc * 10000 + ((hw >> 13) & 7) * 1000 + ((hw >> 8) & 0xf)
g = collections.defaultdict(list)
for i, k in enumerate(cu): g[k].append(end[i])
print("CUs", len(g), "workgroups per CU", collections.Counter(len(v) for v in g.values()), "within-CU spread of end times: median", round(float(np.median([max(v) - min(v) for v in g.values()])), 2))
