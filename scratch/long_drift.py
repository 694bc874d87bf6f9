"""Does the error of the bf16 x 3 kernels grow with time? Ten seconds (cfg3) / five seconds (cfg5) of audio through the pool at
the BASELINE stream counts, a few distinct streams against per-stream oracle plugins, error at checkpoints — once on the split
kernels, once on their fp32 MFMA twins (profiles/r04_long_drift.txt). usage: python scratch/long_drift.py"""
import importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
EQ = dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)
def run(tag, env, mk, S, distinct, seconds, ckw):
    for k in ("AIDAX_GRU_GM", "AIDAX_LP_SPLIT"): os.environ.pop(k, None)
    os.environ.update(env)
    j = W.make_model(**mk); p = W.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json")); spec = O.parse_model(j)
    n, block = int(48000 * seconds) // 256 * 256, 256
    base = W.signal(distinct, n, seed=505)
    idx = (np.arange(S) * 7) % distinct
    first = [int(np.argmax(idx == k)) for k in range(distinct)]
    pool = ax.Pool(S, block); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(**ckw))
    plugs = []
    for _ in range(distinct):
        q = O.OraclePlugin(); q.set_model(O.OracleModel(spec)); plugs.append(q)
    co = O.default_controls(**ckw)
    xd = torch.empty(S, block, device="cuda"); yd = torch.empty_like(xd)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)          # one stream for the copies and the passes: program order
    idx_t = torch.from_numpy(idx).cuda()
    based = torch.from_numpy(base).cuda()
    worst, marks, out = 0.0, [1, 2, 5, 10], []
    t0 = time.time()
    for bi in range(n // block):
        b = bi * block
        xd.copy_(based[idx_t, b:b + block])
        pool.process_device(xd.data_ptr(), yd.data_ptr(), block, st.cuda_stream)
        st.synchronize()
        got = yd[first].cpu().numpy()
        for k in range(distinct):
            worst = max(worst, float(np.abs(got[k] - plugs[k].run(co, base[k, b:b + block])).max()))
        t = (b + block) / 48000.0
        if marks and t >= marks[0]:
            out.append(f"{marks.pop(0)} s: {worst:.2e}")
    print(f"{tag:34s} {pool.kernel_name:10s} max |err| so far at " + ", ".join(out) + f"   ({time.time() - t0:.0f} s)", flush=True)
    pool.close()
g = dict(kind="gru", hidden=64, input_size=3, seed=64)
l = dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2)
run("cfg3 GRU-64, bf16 x 3 (6 products)", {}, g, 4096, 4, 10, EQ)
run("cfg3 GRU-64, fp32 MFMA", {"AIDAX_GRU_GM": "f32"}, g, 4096, 4, 10, EQ)
run("cfg5 LSTM-96 x 2, bf16 x 3", {}, l, 2048, 2, 5, {})
run("cfg5 LSTM-96 x 2, fp32 MFMA", {"AIDAX_LP_SPLIT": "0"}, l, 2048, 2, 5, {})
