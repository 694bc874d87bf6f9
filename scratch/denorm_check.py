# do the fp32 gain smoothers behave like the CPU's through the denormal range? a muted pool (no model: master target 0)
# for 150 000 samples, bit for bit against the oracle
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
S, n, blocks = 3, 256, 600
pool = ax.Pool(S, n)
pool.set_controls(ax.default_controls(master_db=6.0))
pool.activate()
plugs = [O.OraclePlugin() for _ in range(S)]
for p in plugs: p.activate()
x = modelgen.signal(S, n * blocks, seed=3)
first = None
tiny = 0
for b in range(blocks):
    blk = np.ascontiguousarray(x[:, b * n:(b + 1) * n])
    if b == 2:      # un-mute for a while so that the master ramp is up, then mute: it decays through the denormals
        for p in plugs: p.set_loading(False)
        pool.set_loading(False)
    if b == 40:
        for p in plugs: p.set_loading(True)
        pool.set_loading(True)
    got = pool.process(blk)
    want = np.stack([plugs[s].run(O.default_controls(master_db=6.0), blk[s]) for s in range(S)])
    tiny += int(((np.abs(want) > 0) & (np.abs(want) < 1.2e-38)).sum())
    if first is None and not np.array_equal(got, want):
        first = (b, float(np.abs(got - want).max()), float(np.abs(want).max()))
print("denormal output samples seen:", tiny, " first block that is not bit-identical:", first)
