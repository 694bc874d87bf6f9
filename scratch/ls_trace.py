"""Where does a tick of k_mfma_ls go? Needs the measurement build (make LIBDIR=build/lib_tr OBJDIR=build/obj_tr EXTRA=-DAIDAX_LP_TRACE
build/lib_tr/libaidax_hip.so): both workgroups of stream group 0 stamp the shader clock at eight points of ticks 96..103.
usage: AIDAX_LIB=scratch/prev_lib/libaidax_tr.so python scratch/ls_trace.py"""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
tf = os.path.join(tempfile.mkdtemp(), "trace.bin")
os.environ["AIDAX_LP_TRACE_FILE"] = tf
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
mk, S = dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), 2048
j = modelgen.make_model(**mk); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
pool.close()
t = np.fromfile(tf, dtype=np.uint64)[:512].reshape(2, 8, 4, 8).astype(np.int64)      # [role][tick][wave][stamp]
names = ["head: fetch/xin issue, poll", "dense, ship, acc init, frag loads", "phase A", "stash + phase B", "phase C", "tail", "barrier"]
for role, rn in enumerate(("first layer", "last layer")):
    for w in range(4):
        d = np.diff(t[role, :, w, :], axis=1)
        print(f"{rn} wave {w}: tick period {np.diff(t[role, :, w, 0]).mean():7.0f} | " + " | ".join(f"{names[k]} {d[:, k].mean():6.0f}" for k in range(7)))
