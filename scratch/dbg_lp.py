import importlib, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
H = int(sys.argv[1]) if len(sys.argv) > 1 else 48
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sizes = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [256, 1, 5, 700, 16, 3, 255, 40]
p = W.write_model(W.make_model("lstm", H, 1, seed=5, n_rnn=NL), os.path.join(tempfile.mkdtemp(), "m.json"))
x = W.signal(S, sum(sizes), seed=3)
outs = {}
for lp in ("1", "0"):
    os.environ["AIDAX_MFMA_LP"] = lp
    pool = ax.Pool(S, 2048); pool.set_model(ax.Model(p))
    o, pos = [], 0
    for n in sizes:
        o.append(pool.process(np.ascontiguousarray(x[:, pos:pos + n]))); pos += n
    outs[lp] = o; pool.close()
for i, n in enumerate(sizes):
    d = np.abs(outs["1"][i] - outs["0"][i])
    bad = np.argwhere(d > 0)
    print("block", i, "n", n, "max diff %.3e" % d.max(), "first bad frame", bad[:, 1].min() if len(bad) else None,
          "streams", sorted(set(bad[:, 0]))[:10] if len(bad) else None, "count", len(set(bad[:, 0])) if len(bad) else 0)
# state comparison after each block for stream 0 (and 1)
sts = {}
for lp in ("1", "0"):
    os.environ["AIDAX_MFMA_LP"] = lp
    pool = ax.Pool(S, 2048); pool.set_model(ax.Model(p))
    pos = 0; rec = []
    for n in sizes:
        pool.process(np.ascontiguousarray(x[:, pos:pos + n])); pos += n
        rec.append([pool.read_state(stream=0, layer=l, hidden=128) for l in range(NL)])
    sts[lp] = rec; pool.close()
for i, n in enumerate(sizes):
    for l in range(NL):
        dh = np.abs(sts["1"][i][l][0] - sts["0"][i][l][0]); dc = np.abs(sts["1"][i][l][1] - sts["0"][i][l][1])
        print("block", i, "n", n, "layer", l, "h diff units", list(np.nonzero(dh)[0][:12]), "c diff units", list(np.nonzero(dc)[0][:12]))
