# LV2 shell in hub mode (AIDAX_HUB): several plugin instances of one process under the mock host, sharing one pool pass
# per period with the other instances that play the same model file. Random patch:Set requests move instances between
# hubs: like a model swap of the reference (rt-neural-generic.cpp:868-875) the instance keeps its biquad memories and
# gain smoothers (the new seat adopts them on the device, behind the old seat's last block) and the new DynamicModel is
# built around the PARAM targets of the playing one (:822-825) — the mirror is ONE oracle plugin per instance for its
# whole life. Workers and responses come late, controls move. An instance must only ever deliver the oracle's output of
# its previous block, or silence (first block on a seat, a pass too old) — and silence must stay rare.
# usage: python tests/soak_lv2_hub.py [periods]
import os, shutil, sys, tempfile
os.environ["AIDAX_HUB"] = "4"
os.environ["AIDAX_HUB_DEADLINE_US"] = "0"          # a python host keeps no audio clock
os.environ.setdefault("AIDAX_STRICT_REFERENCE_SET", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import lv2host, modelgen

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "9")))
bundle = tempfile.mkdtemp()
src = os.path.join(lv2host.ROOT, "tests", "golden", "models")
dst = os.path.join(bundle, "models", "deer ink studios")
os.makedirs(dst)
files = []
for f in sorted(os.listdir(src))[:2]:
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    files.append(os.path.join(dst, f))
for name, kw in (("gru16.json", dict(kind="gru", hidden=16, input_size=3, seed=7)), ("lstm20.json", dict(kind="lstm", hidden=20, input_size=2, seed=8))):
    files.append(modelgen.write_model(modelgen.make_model(**kw), os.path.join(bundle, "models", name)))
files.append(os.path.join(bundle, "models", "missing.json"))
specs = {f: (O.load_model(f) if os.path.exists(f) else None) for f in files}

N, n = 4, 128
hosts = [lv2host.Host(bundle_dir=bundle, block=n) for _ in range(N)]
assert all(h.handle for h in hosts)


class Mirror:
    def __init__(self):
        self.plug = O.OraclePlugin()
        self.prev = None
        self.seated = False
        self.answers = []                 # specs work() has answered, in the order of the host's response queue
        self.log = []


mir = [Mirror() for _ in range(N)]
controls = lambda h: O.default_controls(**{lv2host.FIELD[k]: h.ctl[k].value for k in lv2host.FIELD})
stats = dict(blocks=0, delivered=0, silent=0, swaps=0, failed=0)
worst = 0.0
for p in range(periods):
    for i, h in enumerate(hosts):
        m = mir[i]
        asked = None
        if rs.rand() < 0.04 or p == i:     # everybody loads something in the first periods
            asked = files[rs.randint(len(files) - 1)] if p == i else files[rs.randint(len(files))]
            h.send_patch_set(asked)
        if rs.rand() < 0.2:
            h.controls(PARAM1=float(rs.rand()), PREGAIN=float(rs.uniform(-6, 3)), MASTER=float(rs.uniform(-9, 3)),
                       BASS=float(rs.uniform(-5, 5)), NETBYPASS=float(rs.rand() > 0.8))
        x = rs.uniform(-0.6, 0.6, size=n).astype(np.float32)
        got = h.run(x)
        stats["blocks"] += 1
        if asked is not None:
            m.plug.set_loading(True)
            m.log.append((p, "asked", os.path.basename(asked)))
        lin = 10.0 ** (max(0.0, h.ctl["MASTER"].value) / 20.0) * 10.0 ** (max(0.0, h.ctl["BASS"].value) / 20.0)
        if m.prev is not None and float(np.abs(got - m.prev).max()) <= 1e-5 * lin:
            stats["delivered"] += 1
            worst = max(worst, float(np.abs(got - m.prev).max()))
        elif not got.any():
            stats["silent"] += 1
        else:
            print("MISMATCH period", p, "instance", i, "err", float(np.abs(got - m.prev).max()) if m.prev is not None else None, "asked", asked)
            print("  got", got[:4], "prev", None if m.prev is None else m.prev[:4])
            for j_, mm in enumerate(mir): print("  instance", j_, mm.log[-8:])
            sys.exit(1)
        want = m.plug.run(controls(h), x)
        m.prev = want if m.plug.model is not None else None      # before the first model the plugin has no seat: nothing is computed
        if h.work_queue and rs.rand() < 0.7:
            for msg in h.work_queue:
                if int.from_bytes(msg[:4], "little") == 0:
                    spec = specs.get(msg[4:].split(b"\0")[0].decode())
                    if spec is not None:
                        # work() reads the playing model's PARAM targets now (:822-825)
                        old = m.plug.model.ptr.contents if m.plug.model is not None else None
                        m.answers.append((spec, old.param1Coeff.target if old else 0.0, old.param2Coeff.target if old else 0.0))
                    else: stats["failed"] += 1
            h.pump_worker()
            m.log.append((p, "pumped", len(m.answers)))
            assert len(h.responses) == len(m.answers)
        if h.responses and rs.rand() < 0.7:
            k = h.deliver_responses()
            for spec, p1, p2 in m.answers[:k]:
                if not m.seated:                                   # the first model: the instance gets its first seat, a stream
                    m.plug = O.OraclePlugin()                      # from instantiate() + activate() on (before it, hub mode keeps none)
                    m.plug.activate()
                    m.seated = True
                m.plug.set_model(O.OracleModel(spec, p1, p2))      # the swap of :868-875: the plugin's own members stay
                m.prev = None                                      # (the hub's one period of latency starts over on the new seat)
                m.log.append((p, "seat change"))
                stats["swaps"] += 1
            del m.answers[:k]
for h in hosts:
    h.pump_worker()
    h.close()
print(stats, "worst", worst)
if stats["silent"] > 0.15 * stats["blocks"]:
    print("TOO MUCH SILENCE"); sys.exit(1)
print("lv2 hub soak ok:", periods, "periods,", stats["delivered"], "blocks delivered,", stats["silent"], "silent,", stats["swaps"], "seat changes")
