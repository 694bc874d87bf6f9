# Soak: random block sizes, control flips, activates and model swaps for many blocks; three streams against the
# oracle's plugin mirror. usage: [SOAK_STREAMS=n] [AIDAX_KERNEL=<form>] python tests/soak.py [blocks]
import importlib, os, sys, tempfile, threading
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "2026")))
d = tempfile.mkdtemp()
models = []
SHAPES = (("lstm", 32, 1, 1), ("gru", 16, 3, 1), ("lstm", 12, 2, 1), ("gru", 64, 1, 1), ("lstm", 32, 2, 2), ("conv", 16, 1, 1))
if os.environ.get("SOAK_MODELS") == "wide":       # the wide one-layer cells and stacks: at ~4200 streams k_lstm_gs, k_mfma_ls1, k_gru_gs, k_mfma_ls (in ranges)
    SHAPES = (("lstm", 64, 1, 1), ("lstm", 80, 2, 1), ("gru", 80, 3, 1), ("lstm", 40, 2, 1), ("gru", 64, 3, 1), ("lstm", 96, 1, 2), ("lstm", 12, 2, 1))
for kind, H, I, L in SHAPES:
    # the last two are extensions — a stacked model (k_mfma_lp) and a conv stack (k_conv_mfma, chain passes inside the
    # launch) — so swaps also cross kernel families
    # the GRU-16 file carries a numeric samplerate: its PARAM smoothers run at the model's rate, not the host's (:1053-1060)
    j = modelgen.make_model(kind, H, I, seed=H + I, n_rnn=L, samplerate=44100.0 if (kind, H) == ("gru", 16) else None,
                            in_skip=1 if (kind, H) == ("lstm", 12) else None, in_gain=-2.0 if H == 12 else None, out_gain=1.5 if H == 12 else None)
    models.append((ax.Model(modelgen.write_model(j, os.path.join(d, f"{kind}{H}x{L}.json"))), O.parse_model(j)))
S, MAXF = int(os.environ.get("SOAK_STREAMS", "70")), int(os.environ.get("SOAK_MAXF", "256"))
if MAXF > 512 and os.environ.get("AIDAX_KERNEL") == "valu":   # the VALU conv kernel keeps a whole block in two LDS planes
    models = [m for m in models if m[1].rnn_type != "conv1d"]      # 70: the resident forms (pipe, lp, fused conv); ~4200: the many-streams forms
SR = float(os.environ.get("SOAK_SR", "48000"))                    # the host's rate: gain smoothers, filter designs
pool = ax.Pool(S, MAXF, SR)
watch = sorted({0, max(0, S // 2 - 2), S - 1})          # (a pool of one or three streams: fewer watched)
plugs = {s: O.OraclePlugin(SR) for s in watch}
cur = 0
pool.set_model(models[cur][0])
for s in watch: plugs[s].set_model(O.OracleModel(models[cur][1]))
kw = {s: {} for s in range(S)}
worst = 0.0
names = set()
for b in range(blocks):
    r = rs.rand()
    if r < 0.03:
        cur = rs.randint(len(models))
        if rs.rand() < 0.5:
            pool.set_model(models[cur][0])
        else:                                        # the two-thread form of a swap: prepare on a worker, commit here
            box = {}
            t = threading.Thread(target=lambda: box.update(sg=pool.prepare_model(models[cur][0])))
            t.start(); t.join()
            pool.commit_model(box["sg"])
            pool.staged_free(box["sg"])
        for s in watch:
            old = plugs[s].model.ptr.contents
            plugs[s].set_model(O.OracleModel(models[cur][1], old.param1Coeff.target, old.param2Coeff.target))
    elif r < 0.06:
        pool.activate()
        for s in watch: plugs[s].activate()
    elif r < 0.35:
        s = watch[rs.randint(len(watch))] if rs.rand() < 0.7 else rs.randint(S)
        choice = rs.randint(7)
        k = dict(kw[s])
        if choice == 0: k["param1"] = float(rs.rand())
        elif choice == 1: k["param2"] = float(rs.rand())
        elif choice == 2: k["enabled"] = float(rs.rand() > 0.2)
        elif choice == 3: k["net_bypass"] = float(rs.rand() > 0.7)
        elif choice == 4: k["eq_position"] = float(rs.rand() > 0.5); k["bass_boost_db"] = float(rs.uniform(-8, 8))
        elif choice == 5: k["pregain_db"] = float(rs.uniform(-12, 12)); k["master_db"] = float(rs.uniform(-15, 15))
        else: k["mid_type"] = float(rs.rand() > 0.5); k["mid_boost_db"] = float(rs.uniform(-8, 8)); k["in_lpf_pc"] = float(rs.choice([0.0, 30.0, 66.216, 100.0]))
        kw[s] = k
        pool.set_controls(ax.default_controls(**k), stream=s)
    n = int(rs.choice([256, 256, 256, 128, 64, 17, 1, 0, 255, 200] + ([MAXF, MAXF - 1, 1000, 513, 300] if MAXF > 256 else [])))
    x = (rs.uniform(-0.6, 0.6, size=(S, n))).astype(np.float32)
    got = pool.process(x)
    names.add(pool.kernel_name)
    for s in watch:
        want = plugs[s].run(O.default_controls(**kw[s]), x[s])
        if n:
            e = float(np.abs(got[s] - want).max())
            worst = max(worst, e)
            if e > 5e-6:
                print("MISMATCH block", b, "stream", s, "n", n, "err", e, "kernel", pool.kernel_name, "controls", kw[s]); sys.exit(1)
print("soak ok:", blocks, "blocks, worst |err| =", worst, "kernels:", sorted(names))
