import importlib, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import torch
B = importlib.import_module("aidadsp-lv2_amd.binding")
if len(sys.argv) > 1 and sys.argv[1] != "0":
    path = os.path.join(os.getcwd(), "scratch", "probe", sys.argv[1], "libaidax_hip.so")
    B.lib_path = lambda: path
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
eq = dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)
def run(label, mkw, S, ckw={}, n=256, steps=40):
    p = modelgen.write_model(modelgen.make_model(**mkw), os.path.join(d, "m.json"))
    pool = ax.Pool(S, n); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ckw))
    x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for _ in range(5): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print(f"variant {sys.argv[1] if len(sys.argv)>1 else 0}: {label:14s} {pool.kernel_name:22s} S={S:6d}: {e0.elapsed_time(e1)/steps*1e3:9.2f} us/step", flush=True)
    pool.close()
run("cfg3 gru64/3", dict(kind="gru", hidden=64, input_size=3, seed=64), 4096, eq)
run("gru32", dict(kind="gru", hidden=32, input_size=1, seed=32), 2048)
run("lstm32 2048", dict(kind="lstm", hidden=32, input_size=1, seed=32), 2048)
run("lstm32 16k", dict(kind="lstm", hidden=32, input_size=1, seed=32), 16384, steps=10)
run("lstm80 4k", dict(kind="lstm", hidden=80, input_size=1, seed=80), 4096, steps=10)
run("cfg5", dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), 2048, steps=10)
run("cfg4", dict(kind="conv", hidden=16, input_size=1, seed=1608), 1024)
run("gru16 1024", dict(kind="gru", hidden=16, input_size=1, seed=16), 1024, steps=200)
run("gru8 1024", dict(kind="gru", hidden=8, input_size=1, seed=8), 1024, steps=200)
