"""Where does a tick of k_gru_gs go? Needs the measurement build (make LIBDIR=build/lib_tr OBJDIR=build/obj_tr
EXTRA=-DAIDAX_LP_TRACE build/lib_tr/libaidax_hip.so): workgroup 0 stamps the shader clock at six points of ticks 96..103 on
every wave and leaves the stamps in its output rows. usage: AIDAX_LIB=build/lib_tr/libaidax_hip.so python scratch/gs_trace.py"""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
mk, S = dict(kind="gru", hidden=64, input_size=3, seed=64), 4096
ctl = dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)
j = modelgen.make_model(**mk); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ctl))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(50): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
raw = y.cpu().numpy().reshape(-1)[:1536].copy()
pool.close()
t = raw.view(np.uint64).reshape(8, 12, 8)[:, :, :6].astype(np.int64)      # [tick][wave][stamp]; waves 4..7: the helpers (stamps 0, 4, 5)
names = ["dense+reads", "fp32+bf16 MFMA issue", "drain+cell", "split+publish", "barrier"]
for w in range(4):
    d = np.diff(t[:, w, :], axis=1)
    print(f"main {w}: tick period {np.diff(t[:, w, 0]).mean():7.0f} | " + " | ".join(f"{names[k]} {d[:, k].mean():6.0f}" for k in range(5)))
for w in range(4, 8):
    print(f"helper {w}: tick period {np.diff(t[:, w, 0]).mean():7.0f} | work per tick {t[:, w, 4] - t[:, w, 0]} | barrier {(t[:, w, 5] - t[:, w, 4]).mean():6.0f}")
print("arrival at the barrier (stamp 4) relative to the earliest, tick 2:", t[2, :8, 4] - t[2, :8, 4].min())
