"""k_conv_st's timeline (cfg4): measurement build  make -s -j8 OBJDIR=build/obj_cv LIBDIR=build/lib_cv EXTRA=-DAIDAX_CONV_TRACE build/lib_cv/libaidax_hip.so
AIDAX_LIB=build/lib_cv/libaidax_hip.so python scratch/st_trace.py
Per wave and tick: when the tick's work began (cycles / 16 from the kernel's start) and how long it took; the rest of a tick is the barrier."""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
w = bench.WORKLOADS["cfg4"]
j = modelgen.make_model(**w["model"]); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
S = int(os.environ.get("ST_STREAMS", w["streams"]))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(200): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name, S, "streams")
raw = y[:, :168].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
ghz = 2.4
T = int(raw[0, 161])
begin = ((raw[:, :128] >> 16) * 16).reshape(S, 4, 32)[:, :, :T] / ghz / 1e3      # us
work = (raw[:, :128] & 0xffff).reshape(S, 4, 32)[:, :, :T] / ghz / 1e3
total = raw[:, 160] / ghz / 1e3
print(f"kernel (chain wave, start -> row stored): median {np.median(total):.2f} us, p10 {np.percentile(total, 10):.2f}, p90 {np.percentile(total, 90):.2f}, max {total.max():.2f}")
print("tick:   begin (median us) | work per wave 0..3 (median us)")
for t in range(min(T, 32)):
    print(f"  {t:2d}   {np.median(begin[:, 0, t]):6.2f}   | " + "  ".join(f"{np.median(work[:, k, t]):5.2f}" for k in range(4)))
d = np.diff(np.median(begin[:, 0, :], axis=0))
print(f"tick length: median {np.median(d):.2f} us, ticks 0..3 {d[:4].round(2)}, steady (8..20) {np.median(d[8:20]):.2f}, last {d[-3:].round(2)}")
print(f"first tick begins at {np.median(begin[:, 0, 0]):.2f} us; work sums per wave (us): " + "  ".join(f"{np.median(work[:, k, :].sum(axis=1)):5.2f}" for k in range(4)))

w0, w1 = raw[:, 162], raw[:, 163]                    # s_memrealtime, 100 MHz, one counter for the chip
print(f"wall clock: first start -> last start {(w0.max() - w0.min()) / 100:6.2f} us, first start -> last end {(w1.max() - w0.min()) / 100:6.2f} us, "
      f"per workgroup start -> end median {np.median(w1 - w0) / 100:6.2f} max {(w1 - w0).max() / 100:6.2f}")
st = (w0 - w0.min()) / 100
print("start times (us) histogram:", np.histogram(st, bins=[0, 1, 2, 5, 10, 20, 30, 40, 50, 100])[0].tolist())
hw, xcc = raw[:, 164], raw[:, 165] & 15
cu = (xcc << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15)
per = np.bincount(np.unique(cu, return_inverse=True)[1])
print("workgroups per (xcc, se, cu):", np.bincount(per).tolist(), " distinct CUs", len(per))
late = st > 5
print(f"late starters: {late.sum()}; their kernels {np.median(total[late]) if late.any() else 0:.2f} us; early ones {np.median(total[~late]):.2f}")

simd = (raw[:, 128:132] >> 4) & 3                     # HW_ID[5:4] of the waves that ran roles 0 .. 3
print("SIMD of role 0..3, first 8 workgroups:", simd[:8].tolist(), " ... workgroups 256..259:", simd[256:260].tolist())
print("roles per SIMD over all workgroups (rows: role, cols: SIMD):", [np.bincount(simd[:, r], minlength=4).tolist() for r in range(4)])
bid = raw[:, 166]
for c in np.unique(cu)[:3]:
    print("  block ids on one CU:", sorted(bid[cu == c].tolist()))
