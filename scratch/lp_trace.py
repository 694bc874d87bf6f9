"""Where does a tick of k_mfma_lp go? Needs the measurement build (make LIBDIR=build/trace/lib OBJDIR=build/trace/obj
EXTRA=-DAIDAX_LP_TRACE build/trace/lib/libaidax_hip.so, copied to scratch/prev_lib/libaidax_trace.so): workgroup 0 stamps the
shader clock at six points of ticks 96..103 on every wave. usage: AIDAX_LIB=scratch/prev_lib/libaidax_trace.so python scratch/lp_trace.py [cfg3|cfg5]"""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
tf = os.path.join(tempfile.mkdtemp(), "trace.bin")
os.environ["AIDAX_LP_TRACE_FILE"] = tf
os.environ["AIDAX_KERNEL"] = "mfma"; os.environ["AIDAX_MFMA_LP"] = "1"
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
if which == "cfg3":
    mk, S = dict(kind="gru", hidden=64, input_size=3, seed=64), 4096
else:
    mk, S = dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), 2048
j = modelgen.make_model(**mk); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
pool.close()
t = np.fromfile(tf, dtype=np.uint64).reshape(8, 12, 8)[:, :, :6].astype(np.int64)      # [tick][wave][stamp]; waves 8..11: the helpers (stamps 0, 4, 5)
names = ["start->pre", "pre->gates", "gates issue", "cell update", "barrier"]
for w in range(8):
    d = np.diff(t[:, w, :], axis=1)
    period = np.diff(t[:, w, 0])
    print(f"wave {w}: tick period {period.mean():7.0f} | " + " | ".join(f"{names[k]} {d[:, k].mean():6.0f}" for k in range(5)))
nw = 12 if t[2, 8, 0] else 8
for w in range(8, nw):
    print(f"helper {w}: tick period {np.diff(t[:, w, 0]).mean():7.0f} | work {(t[:, w, 4] - t[:, w, 0]).mean():6.0f} (per tick: {t[:, w, 4] - t[:, w, 0]}) | barrier {(t[:, w, 5] - t[:, w, 4]).mean():6.0f}")
print("arrival at the barrier (stamp 4) relative to the earliest, tick 2:", t[2, :nw, 4] - t[2, :nw, 4].min())
