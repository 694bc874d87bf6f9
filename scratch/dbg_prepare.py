# tests/test_gpu_parity.py::test_worker_thread_prepares_while_the_audio_thread_processes with a log of what failed where
# (taskset -c 0,1 python scratch/dbg_prepare.py reproduces the failure that some boxes show)
import importlib, os, sys, tempfile, queue, threading
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import modelgen
from tests.test_gpu_parity import _model_file, _ctl_pair
import pathlib
ax = importlib.import_module("aidadsp-lv2_amd")
from oracle import oracle as O
tmp = pathlib.Path(tempfile.mkdtemp())
kinds = [dict(kind="lstm", hidden=16, input_size=1, seed=51), dict(kind="gru", hidden=24, input_size=2, seed=52),
         dict(kind="lstm", hidden=32, input_size=1, seed=53, n_rnn=2), dict(kind="conv", hidden=16, input_size=1, seed=54)]
files = [_model_file(tmp, f"m{i}", **kw) for i, kw in enumerate(kinds)]
models = [ax.Model(p) for p, _ in files]
S, n, blocks = 6, 128, 200
x = modelgen.signal(S, n * blocks, seed=71)
cg, co = _ctl_pair(param1=0.35, pregain_db=2.0, bass_boost_db=3.0)
pool = ax.Pool(S, n); pool.set_model(models[0]); pool.set_controls(cg)
plugs = [O.OraclePlugin() for _ in range(S)]
for p in plugs: p.set_model(O.OracleModel(files[0][1]))
ready, retired = queue.Queue(maxsize=1), queue.Queue()
stop = threading.Event()
def worker():
    k = 0
    while not stop.is_set():
        while not retired.empty(): pool.staged_free(retired.get())
        mi = (1, 2, 3, 0, 2, 1, 3)[k % 7]
        sg = pool.prepare_model(models[mi]); k += 1
        while not stop.is_set():
            try: ready.put((mi, sg), timeout=0.01); break
            except queue.Full: pass
        else: pool.staged_free(sg)
    while not retired.empty(): pool.staged_free(retired.get())
t = threading.Thread(target=worker); t.start()
cur, swaps, since = 0, 0, 0
try:
    for b in range(blocks):
        try: mi, sg = ready.get_nowait()
        except queue.Empty: mi = None
        if mi is not None:
            pool.commit_model(sg); retired.put(sg); swaps += 1; cur = mi; since = 0
            for p in plugs:
                old = p.model.ptr.contents
                p.set_model(O.OracleModel(files[mi][1], old.param1Coeff.target, old.param2Coeff.target))
        blk = np.ascontiguousarray(x[:, b * n:(b + 1) * n])
        got = pool.process(blk)
        errs = [float(np.abs(got[s_] - plugs[s_].run(co, blk[s_])).max()) for s_ in range(S)]
        if max(errs) > 2e-6:
            print("block", b, "model", cur, kinds[cur]["kind"], kinds[cur]["hidden"], "swaps", swaps, "blocks since swap", since, pool.kernel_name, "errs", ["%.2e" % e for e in errs], flush=True)
            if since > 3: break
        since += 1
finally:
    stop.set(); t.join()
    while not ready.empty(): pool.staged_free(ready.get()[1])
print("done: swaps", swaps)
