"""Where does a tick of k_mfma_ls1 go (LSTM-64, 4096 streams)? Needs the AIDAX_LP_TRACE build: scratch/mkvariant.sh tr -DAIDAX_LP_TRACE;
usage: AIDAX_LIB=scratch/prev_lib/libaidax_tr.so python scratch/ls1_trace.py [kind hidden]"""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
tf = os.path.join(tempfile.mkdtemp(), "trace.bin")
os.environ["AIDAX_LP_TRACE_FILE"] = tf
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"; hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mk, S = dict(kind=kind, hidden=hidden, input_size=1, seed=hidden), 4096
j = modelgen.make_model(**mk); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(20): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(kind, hidden, pool.kernel_name)
pool.close()
t = np.fromfile(tf, dtype=np.uint64)[:512].reshape(2, 8, 4, 8).astype(np.int64)      # [role][tick][wave][stamp]; a lone layer stamps as "last"
names = ["head: xin", "dense, acc init, frag loads", "phase A", "phase B", "rest of the cell update", "tail", "barrier"]
for w in range(4):
    d = np.diff(t[1, :, w, :], axis=1)
    print(f"wave {w}: tick period {np.diff(t[1, :, w, 0]).mean():7.0f} | " + " | ".join(f"{names[k]} {d[:, k].mean():6.0f}" for k in range(7)))
