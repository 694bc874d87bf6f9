# Hub soak: a sequential host with a random life — instances attach and detach, get skipped for a period, change
# controls, the block size changes (also in the middle of a period), the hub's model is swapped — every instance
# against its own oracle plugin, one period late. The mirror below restates the hub's contract (include/aidax.h, "hub"):
# a pass is launched when every attached instance has submitted, when an instance comes back before that, when the
# block size changes, or on flush; an instance reads the output of the pass that carried its previous block if that is
# at most two passes old and has the same length, else silence; an instance that is not part of a pass does not move.
# usage: python tests/soak_hub.py [periods]
import importlib, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 400
NO = set(os.environ.get("SOAK_NO", "").split(","))          # bisecting a failure: skip,nchange,detach,swap,twice,zero
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "77")))
d = tempfile.mkdtemp()
models = []
for kind, H, I, L in (("lstm", 16, 2, 1), ("gru", 24, 1, 1), ("lstm", 12, 3, 1), ("conv", 16, 1, 1), ("lstm", 32, 1, 2)):
    j = modelgen.make_model(kind, H, I, seed=10 * H + I, n_rnn=L)
    models.append((ax.Model(modelgen.write_model(j, os.path.join(d, f"{kind}{H}x{L}.json"))), O.parse_model(j)))

CAP, MAXF = 12, 256
hub = ax.Hub(CAP, MAXF)
hub.set_deadline_us(0)                      # deterministic: the host below decides when a period closes
cur = int(os.environ.get("SOAK_MODEL", "0"))
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
        self.log = []                       # what happened to it lately (printed with a mismatch)
        self.hist = []                      # (model index, controls, block) of every pass it took part in: replayed on a mismatch


glog = []                                   # everything that happened, in order (tail printed with a mismatch)
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
    glog.append(f"pass {launches}: n={period_n} model={cur} slots={list(submitted)} attached={sorted(insts)}")
    for s in submitted:
        it = insts[s]
        it.out = it.plug.run(O.default_controls(**it.kw), it.inbuf)
        it.hist.append([cur, dict(it.kw), it.inbuf.copy(), launches, None])
        it.last_pass = launches
        it.log.append(f"pass {launches}: n={it.inbuf.size} model={cur} kw={it.kw}")
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
    if n and it.last_pass and launches < it.last_pass + 3 and it.out is not None and it.out.size == n:
        want = it.out
    submitted.append(slot)
    it.inbuf = x.copy()
    got = hub.run(slot, x)
    if n and want is it.out and it.hist:
        it.hist[-1][4] = got.copy()          # what the hub delivered for the last pass of its history
    if n:
        e = float(np.abs(got - want).max())
        worst = max(worst, e)
        checked += 1
        if e > 5e-6:
            print("MISMATCH period", p, "slot", slot, "n", n, "err", e, "launches", launches, "last_pass", it.last_pass, "controls", it.kw)
            print("  first bad sample", int(np.argmax(np.abs(got - want) > 5e-6)), "got", got[:4], "want", want[:4])
            for line in it.log[-12:]: print("  ", line)
            # whose fault? the instance's own history through a one-stream pool of its own and through a fresh oracle plugin
            if all(h[0] == it.hist[0][0] for h in it.hist):
                mi = it.hist[0][0]
                pool1 = ax.Pool(1, MAXF); pool1.set_model(models[mi][0]); pool1.activate()
                pl = O.OraclePlugin(); pl.set_model(O.OracleModel(models[mi][1])); pl.activate()
                first_bad = None
                for hi, (_, kw_h, x_h, pass_h, got_h) in enumerate(it.hist):
                    pool1.set_controls(ax.default_controls(**kw_h))
                    g1 = pool1.process(x_h[None, :])[0]
                    w1 = pl.run(O.default_controls(**kw_h), x_h)
                    if got_h is not None and g1.size and first_bad is None and float(np.abs(got_h - g1).max()) > 5e-6:
                        first_bad = (hi, pass_h, x_h.size, float(np.abs(got_h - g1).max()), float(np.abs(g1 - w1).max()))
                print("  replay: first delivered block that differs from a one-stream pool: (history index, pass, n, |hub - pool1|, |pool1 - oracle|) =", first_bad,
                      "of", len(it.hist), "kernel", pool1.kernel_name)
                if first_bad:
                    lo = max(0, first_bad[0] - 3)
                    for hi in range(lo, first_bad[0] + 1):
                        print("    history", hi, "pass", it.hist[hi][3], "n", it.hist[hi][2].size, "checked" if it.hist[hi][4] is not None else "not read back", it.hist[hi][1])
                    for line in glog:
                        if line.startswith("pass ") and it.hist[lo][3] <= int(line.split()[1].rstrip(":")) <= first_bad[1]: print("   ", line[:200])
            print("  --- global")
            for line in glog[-40:]: print("  ", line)
            sys.exit(1)


for p in range(periods):
    # ---- between periods: the population and the model change
    r = rs.rand()
    free = [s for s in range(CAP) if s not in insts]
    if (r < 0.10 or len(insts) < 2 or "attach" in NO) and free:
        slot = hub.attach()
        assert slot == min(free), (slot, free)
        insts[slot] = Inst(slot, models[cur][1])
        insts[slot].log.append(f"period {p}: attached, model {cur}")
        glog.append(f"period {p}: attach slot {slot}")
    elif r < 0.15 and len(insts) > 2 and "detach" not in NO:
        slot = list(insts)[rs.randint(len(insts))]
        hub.detach(slot)
        del insts[slot]
        glog.append(f"period {p}: detach slot {slot}")
    elif r < 0.19 and "swap" not in NO:
        time.sleep(0.02)                    # the last pass has finished: the new model inherits settled PARAM targets
        cur = rs.randint(len(models))
        hub.set_model(models[cur][0])
        glog.append(f"period {p}: hub model -> {cur}")
        for it in insts.values():
            old = it.plug.model.ptr.contents
            it.plug.set_model(O.OracleModel(models[cur][1], old.param1Coeff.target, old.param2Coeff.target))
            it.log.append(f"period {p}: hub model -> {cur}, inherits ({old.param1Coeff.target}, {old.param2Coeff.target})")
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
            glog.append(f"period {p}: slot {it.slot} controls {k}")
    # ---- the period: every instance once (some skipped), now and then one of them with another block size
    n = int(rs.choice([256, 256, 128, 64, 33, 1, 0]))
    if n == 0 and "zero" in NO: n = 64
    if n == 33 and "n33" in NO: n = 32
    if n in (33, 1) and "odd" in NO: n = 32
    order = list(insts)
    rs.shuffle(order)
    for s in order:
        if rs.rand() < 0.07 and "skip" not in NO:
            continue                        # the host skips this instance this period
        ni = n if rs.rand() > 0.04 or "nchange" in NO else int(rs.choice([256, 16 if "odd" in NO else 17]))
        run(s, rs.uniform(-0.6, 0.6, size=ni).astype(np.float32))
        if rs.rand() < 0.05 and "midctl" not in NO:
            # a control moves while the instance's block still waits for its pass: the block was played under the old
            # value, so the hub closes the period first
            it = insts[s]
            new_kw = dict(it.kw, pregain_db=float(rs.uniform(-9, 9)), param1=float(rs.rand()))
            if s in submitted:
                mirror_flush()                                  # ... under the controls in force so far
            it.kw = new_kw
            hub.set_controls(s, ax.default_controls(**it.kw))
            glog.append(f"period {p}: slot {s} controls (block pending) {it.kw}")
    if rs.rand() < 0.1 and order and "twice" not in NO:
        run(order[0], rs.uniform(-0.6, 0.6, size=n).astype(np.float32))      # the same instance twice: closes the period itself
    hub.flush()                             # the launcher thread may not have got to it yet: close the period here
    mirror_flush()
    assert hub.launches == launches, (hub.launches, launches)
print("hub soak ok:", periods, "periods,", checked, "blocks checked,", launches, "passes, worst |err| =", worst)
