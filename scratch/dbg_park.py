# hub-soak failure hunt: a pool of 12 streams, stacked model, random subsets of streams disabled per block, odd block sizes
import importlib, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
j = modelgen.make_model("lstm", 32, 1, seed=321, n_rnn=2)
m, spec = ax.Model(modelgen.write_model(j, os.path.join(d, "m.json"))), O.parse_model(j)
S = 12
for seed in range(6):
    rs = np.random.RandomState(seed)
    pool = ax.Pool(S, 256); pool.set_model(m)
    plugs = [O.OraclePlugin() for _ in range(S)]
    for p in plugs: p.set_model(O.OracleModel(spec)); p.activate()
    pool.activate()
    base = [dict(net_bypass=float(rs.rand() > 0.5), pregain_db=float(rs.uniform(-6, 6)), master_db=float(rs.uniform(-9, 3)),
                 bass_boost_db=float(rs.uniform(-5, 5))) for _ in range(S)]
    bad = None
    for b in range(300):
        n = int(rs.choice([256, 128, 33, 17, 1, 0, 64]))
        on = rs.rand(S) > float(os.environ.get("POFF", "0.5"))
        x = rs.uniform(-0.6, 0.6, size=(S, n)).astype(np.float32)
        for s in range(S):
            pool.set_controls(ax.default_controls(enabled=float(on[s]), **base[s]), stream=s)
        got = pool.process(x)
        for s in range(S):
            if not on[s] and os.environ.get("PARK_NO_CALL"):
                continue                     # the hub's mirror: a parked instance is not called at all
            want = plugs[s].run(O.default_controls(enabled=float(on[s]), **base[s]), x[s])
            if n and np.abs(got[s] - want).max() > 5e-6 and bad is None:
                bad = (b, s, n, float(np.abs(got[s] - want).max()), bool(on[s]))
    print("seed", seed, pool.kernel_name, "first mismatch:", bad, flush=True)
    pool.close()
