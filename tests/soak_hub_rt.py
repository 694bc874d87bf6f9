# Hub soak under real-time pacing: several host threads, each calling its instances once per period with jitter and
# the occasional skipped period, the hub's launcher thread closing periods by DEADLINE (default: half a period) — so
# which pass an instance's block travels in is up to the clock. What does not depend on it: a stream only moves when its
# instance submits, so the k-th run() of an instance returns either the oracle's output for its (k-1)-th block or
# silence (its previous pass too old / first block), never anything else — and silence must stay rare.
# usage: python tests/soak_hub_rt.py [periods]
import importlib, os, sys, tempfile, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")

periods = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(os.environ.get("SOAK_SEED", "3"))
N, T, n = 12, 3, 96                        # instances, host threads, frames per period (2 ms at 48 kHz)
period_s = n / 48000.0
d = tempfile.mkdtemp()
kind = os.environ.get("SOAK_RT_MODEL", "lstm12")
j = modelgen.make_model("lstm", 12, 2, seed=5) if kind == "lstm12" else modelgen.make_model("lstm", 32, 1, seed=6, n_rnn=2)
m, spec = ax.Model(modelgen.write_model(j, os.path.join(d, "m.json"))), O.parse_model(j)
hub = ax.Hub(N, 128)
hub.set_model(m)
if "SOAK_RT_DEADLINE_US" in os.environ:
    hub.set_deadline_us(int(os.environ["SOAK_RT_DEADLINE_US"]))
slots = [hub.attach() for _ in range(N)]
# the oracle plugins are built here, one after another: ModelSpec.descs() keeps the weight buffers of the LAST call
# alive only, so building models from one spec on several threads at once would hand the C side freed memory
all_plugs = []
for i in range(N):
    pl = O.OraclePlugin()
    pl.set_model(O.OracleModel(spec)); pl.activate()
    all_plugs.append(pl)
stats = dict(blocks=0, delivered=0, silent=0, bad=0, worst=0.0)
lock = threading.Lock()
errors = []
start = time.perf_counter() + 0.05


def host_thread(t):
    rs = np.random.RandomState(seed * 100 + t)
    mine = [i for i in range(N) if i % T == t]
    plugs = {i: all_plugs[i] for i in mine}
    kw = {i: dict(param1=float(rs.rand()), pregain_db=float(rs.uniform(-6, 6))) for i in mine}
    for i in mine:
        hub.set_controls(slots[i], ax.default_controls(**kw[i]))
    prev = {i: None for i in mine}
    loc = dict(blocks=0, delivered=0, silent=0, bad=0, worst=0.0)
    try:
        for p in range(periods):
            # wait for the period's start, plus this thread's jitter
            target = start + p * period_s + rs.uniform(0, 0.3) * period_s
            while time.perf_counter() < target:
                time.sleep(0.0001)
            for i in mine:
                if rs.rand() < 0.03:
                    continue                                   # this instance is not called this period
                if rs.rand() < 0.05:
                    time.sleep(rs.uniform(0, 0.6) * period_s)   # the host dawdles: the deadline may pass in between
                x = rs.uniform(-0.6, 0.6, size=n).astype(np.float32)
                got = hub.run(slots[i], x)
                loc["blocks"] += 1
                if prev[i] is not None:
                    e = float(np.abs(got - prev[i]).max())
                    if e < 5e-6:
                        loc["delivered"] += 1; loc["worst"] = max(loc["worst"], e)
                    elif not got.any():
                        loc["silent"] += 1
                    else:
                        loc["bad"] += 1
                        errors.append((t, p, i, e))
                prev[i] = plugs[i].run(O.default_controls(**kw[i]), x)
    except Exception as ex:                                      # pragma: no cover
        errors.append((t, "exception", repr(ex)))
    with lock:
        for k in ("blocks", "delivered", "silent", "bad"):
            stats[k] += loc[k]
        stats["worst"] = max(stats["worst"], loc["worst"])


threads = [threading.Thread(target=host_thread, args=(t,)) for t in range(T)]
for th in threads: th.start()
for th in threads: th.join()
hub.flush()
print("launches", hub.launches, "by deadline", hub.deadline_launches, stats)
if errors or stats["bad"]:
    print("MISMATCH", errors[:5]); sys.exit(1)
# Silence is what a late pass costs by design; how much of it a run sees is the HOST's doing (three paced threads + the launcher
# need cores of their own): the 5 % bound holds where the process has at least eight, on fewer (a pinned or crowded box) the
# contract checked is the audio alone — every delivered block the oracle's, none wrong.
cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
if stats["silent"] > 0.05 * stats["blocks"]:
    if cores >= 8 and not os.environ.get("SOAK_RT_ANY_SILENCE"):
        print("TOO MUCH SILENCE", stats); sys.exit(1)
    print("(silence above 5 % on", cores, "cores: not judged)")
print("hub rt soak ok:", periods, "periods,", stats["delivered"], "blocks delivered,", stats["silent"], "silent, worst |err| =", stats["worst"])
