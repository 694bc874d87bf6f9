"""k_conv_ms2 (cfg4, full blocks): where a workgroup's time goes, by role. Measurement build: make OBJDIR=build/obj_cv LIBDIR=build/lib_cv
EXTRA="-DAIDAX_TEST_HOOKS -DAIDAX_CONV_TRACE" build/lib_cv/libaidax_hip.so; AIDAX_LIB=build/lib_cv/libaidax_hip.so python scratch/conv_trace2.py"""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
w = bench.WORKLOADS["cfg4"]
j = modelgen.make_model(**w["model"]); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
S = w["streams"]
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(200): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
raw = y[:, :24].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
ghz = 2.4
names = {1: "chain: pre(half 0) done", 2: "chain: arrives at the end of half 0's barriers", 3: "chain: pre(half 1) done", 4: "chain: past the half barrier",
         5: "chain: arrives at the end of half 1's barriers", 6: "chain: past the last barrier", 7: "chain: post passes done, stored",
         8: "compute: past B0", 9: "compute: half 0 done", 10: "compute: past the half barrier", 11: "compute: half 1 done"}
for k in (1, 8, 2, 3, 9, 4, 10, 5, 11, 6, 7):
    v = raw[:, k] / ghz / 1e3
    print(f"{names[k]:52s} median {np.median(v):7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}")
print("chain macro-steps per launch:", int(np.median(raw[:, 13])))
ph = raw[:, 16:23]
pn = ["history out (12 LDS reads, 12 stores)", "k-loop (9 reads, 36 MFMAs)", "prefix + fragment fetches issued", "wait at the mid barrier", "activation, split, plane writes (3 tiles)", "prefix -> plane, A fragments"]
dd = np.diff(ph, axis=1)
for k in range(6):
    print(f"  layer 4, half 0, compute wave 1: {pn[k]:44s} median {np.median(dd[:, k]) / ghz:8.0f} ns   p10 {np.percentile(dd[:, k], 10) / ghz:8.0f}   p90 {np.percentile(dd[:, k], 90) / ghz:8.0f}")
