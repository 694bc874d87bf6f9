"""Where do the 67 us of the fused conv launch (cfg4) go? Measurement build: make OBJDIR=build/obj_cv LIBDIR=build/lib_cv
EXTRA=-DAIDAX_CONV_TRACE build/lib_cv/libaidax_hip.so; AIDAX_LIB=build/lib_cv/libaidax_hip.so python scratch/conv_trace.py
Every workgroup's chain wave leaves shader-clock stamps in the first floats of its output row."""
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
np.save("gpurun_out/conv_trace.npy", raw)
t = raw[:, :14]
names = ["start", "ctl/state in", "row + coefs in (loop starts)", "pre loop done", "prologue done", "layer-0 prep (barrier, staging)", "layer 0", "layers 1-3",
         "layers 4-7", "dense + barrier", "post coefs in (loop starts)", "post loop done", "stores done"]
d = np.diff(t[:, :13], axis=1)
ghz = 2.4
for k in range(12):
    print(f"{names[k + 1]:34s} median {np.median(d[:, k]) / ghz / 1e3:6.2f} us   p10 {np.percentile(d[:, k], 10) / ghz / 1e3:6.2f}   p90 {np.percentile(d[:, k], 90) / ghz / 1e3:6.2f}")
print(f"whole kernel (chain wave)          median {np.median(t[:, 12]) / ghz / 1e3:6.2f} us   max {t[:, 12].max() / ghz / 1e3:6.2f}")
w0, w1 = raw[:, 13], raw[:, 14]                      # s_memrealtime, 100 MHz, one counter for the chip
print(f"wall clock: first start -> last start {(w0.max() - w0.min()) / 100:6.2f} us, first start -> last end {(w1.max() - w0.min()) / 100:6.2f} us, "
      f"per workgroup start -> end median {np.median(w1 - w0) / 100:6.2f} max {(w1 - w0).max() / 100:6.2f}")

if "conv_ms" in pool.kernel_name:                    # round 5: k_conv_ms stamps the phases of layer 4 on the chain wave (slots 16..21)
    ph = raw[:, 16:22]
    pn = ["MFMA phase (12 reads, 48 MFMAs)", "next layer's fetches + save_history", "wait at the mid barrier", "activation, split, plane writes", "history prefix -> plane"]
    dd = np.diff(ph, axis=1)
    for k in range(5):
        print(f"  layer 4: {pn[k]:40s} median {np.median(dd[:, k]) / ghz:8.0f} ns   p10 {np.percentile(dd[:, k], 10) / ghz:8.0f}   p90 {np.percentile(dd[:, k], 90) / ghz:8.0f}")
