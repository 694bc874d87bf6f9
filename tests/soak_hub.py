# Hub soak: a sequential host with a random life — instances attach and detach, get skipped for a period, change
# controls, the block size changes (also in the middle of a period), the hub's model is swapped — every instance
# against its own oracle plugin, one period late. The mirror below restates the hub's contract (include/aidax.h, "hub"):
# a pass is launched when every attached instance has submitted, when an instance comes back before that, when the
# block size changes, or on flush; an instance reads the output of the pass that carried its previous block if that is
# at most three passes old and has the same length, else silence; an instance that is not part of a pass does not move.
# usage: python tests/soak_hub.py [periods]
import importlib, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "77")))
d = tempfile.mkdtemp()
models = []
for kind, H, I, L in (("lstm", 16, 2, 1), ("gru", 24, 1, 1), ("lstm", 12, 3, 1), ("conv", 16, 1, 1), ("lstm", 32, 1, 2)):
    j = modelgen.make_model(kind, H, I, seed=10 * H + I, n_rnn=L)
    models.append((ax.Model(modelgen.write_model(j, os.path.join(d, f"{kind}{H}x{L}.json"))), O.parse_model(j)))

CAP, MAXF = 12, 256
hub = ax.Hub(CAP, MAXF)
hub.set_deadline_us(0)                      # deterministic: the host below decides when a period closes
cur = 0
hub.set_model(models[cur][0])


class Inst:
    def __init__(self, slot, spec):
        self.slot = slot
        self.plug = O.OraclePlugin()
        self.plug.set_model(O.OracleModel(spec))
        self.plug.activate()
        self.kw = {}
        self.last_pass = 0
        self.out = None                     # what the pass `last_pass` produced for it
        self.inbuf = None


insts = {}                                  # slot -> Inst
submitted = []                              # slots of the period being collected, in order
period_n = 0
launches = 0
worst = 0.0
checked = 0


def mirror_flush():
    global launches
    if not submitted:
        return
    launches += 1
    for s in submitted:
        it = insts[s]
        it.out = it.plug.run(O.default_controls(**it.kw), it.inbuf)
        it.last_pass = launches
    submitted.clear()


def run(slot, x):
    """hub.run + the mirror's expectation for it"""
    global period_n, worst, checked
    it = insts[slot]
    n = x.size
    if slot in submitted or (submitted and n != period_n):
        mirror_flush()
    if not submitted:
        period_n = n
    want = np.zeros(n, np.float32)
    if n and it.last_pass and launches < it.last_pass + 4 and it.out is not None and it.out.size == n:
        want = it.out
    submitted.append(slot)
    it.inbuf = x.copy()
    got = hub.run(slot, x)
    if n:
        e = float(np.abs(got - want).max())
        worst = max(worst, e)
        checked += 1
        if e > 5e-6:
            print("MISMATCH period", p, "slot", slot, "n", n, "err", e, "launches", launches, "last_pass", it.last_pass, "controls", it.kw)
            sys.exit(1)


for p in range(periods):
    # ---- between periods: the population and the model change
    r = rs.rand()
    free = [s for s in range(CAP) if s not in insts]
    if (r < 0.10 or len(insts) < 2) and free:
        slot = hub.attach()
        assert slot == min(free), (slot, free)
        insts[slot] = Inst(slot, models[cur][1])
    elif r < 0.15 and len(insts) > 2:
        slot = list(insts)[rs.randint(len(insts))]
        hub.detach(slot)
        del insts[slot]
    elif r < 0.19:
        time.sleep(0.02)                    # the last pass has finished: the new model inherits settled PARAM targets
        cur = rs.randint(len(models))
        hub.set_model(models[cur][0])
        for it in insts.values():
            old = it.plug.model.ptr.contents
            it.plug.set_model(O.OracleModel(models[cur][1], old.param1Coeff.target, old.param2Coeff.target))
    for it in insts.values():
        if rs.rand() < 0.15:
            c = rs.randint(5)
            k = dict(it.kw)
            if c == 0: k["param1"] = float(rs.rand()); k["param2"] = float(rs.rand())
            elif c == 1: k["enabled"] = float(rs.rand() > 0.3)
            elif c == 2: k["net_bypass"] = float(rs.rand() > 0.6)
            elif c == 3: k["eq_position"] = float(rs.rand() > 0.5); k["bass_boost_db"] = float(rs.uniform(-6, 6)); k["mid_type"] = float(rs.rand() > 0.5)
            else: k["pregain_db"] = float(rs.uniform(-9, 9)); k["master_db"] = float(rs.uniform(-12, 6))
            it.kw = k
            hub.set_controls(it.slot, ax.default_controls(**k))
    # ---- the period: every instance once (some skipped), now and then one of them with another block size
    n = int(rs.choice([256, 256, 128, 64, 33, 1, 0]))
    order = list(insts)
    rs.shuffle(order)
    for s in order:
        if rs.rand() < 0.07:
            continue                        # the host skips this instance this period
        ni = n if rs.rand() > 0.04 else int(rs.choice([256, 17]))
        run(s, rs.uniform(-0.6, 0.6, size=ni).astype(np.float32))
    if rs.rand() < 0.1 and order:
        run(order[0], rs.uniform(-0.6, 0.6, size=n).astype(np.float32))      # the same instance twice: closes the period itself
    hub.flush()                             # the launcher thread may not have got to it yet: close the period here
    mirror_flush()
    assert hub.launches == launches, (hub.launches, launches)
print("hub soak ok:", periods, "periods,", checked, "blocks checked,", launches, "passes, worst |err| =", worst)
