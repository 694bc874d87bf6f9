# LV2 shell soak: the mock host (tests/lv2host.py) with a random life against the oracle's plugin mirror — control
# ports move, patch:Set and state restores ask for models (good ones, the two extension families, a missing file, a
# broken file) while earlier requests are still in flight, the worker runs late or early, responses are delivered
# blocks later, activate(), block sizes 0..512. What must hold (rt-neural-generic.cpp:524-586, :807-893): a patch:Set
# mutes from the run() that sees it; work() loads off the audio thread and answers only on success; work_response()
# swaps, un-mutes and echoes the path; a failed load answers nothing (a request that came by patch:Set stays muted).
# usage: python tests/soak_lv2.py [blocks]
import os, shutil, sys, tempfile
os.environ.setdefault("AIDAX_STRICT_REFERENCE_SET", "0")      # let the extension models through the shell as well
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import lv2host, modelgen

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "5")))
bundle = tempfile.mkdtemp()
src = os.path.join(lv2host.ROOT, "tests", "golden", "models")
dst = os.path.join(bundle, "models", "deer ink studios")
os.makedirs(dst)
rel = []
for f in sorted(os.listdir(src))[:3]:
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    rel.append("models/deer ink studios/" + f)
for name, kw in (("gru16.json", dict(kind="gru", hidden=16, input_size=3, seed=7)),
                 ("lstm20.json", dict(kind="lstm", hidden=20, input_size=2, seed=8, in_skip=1, out_gain=-3.0)),
                 ("lstm32x2.json", dict(kind="lstm", hidden=32, input_size=1, seed=9, n_rnn=2)),
                 ("conv16.json", dict(kind="conv", hidden=16, input_size=1, seed=10))):
    modelgen.write_model(modelgen.make_model(**kw), os.path.join(bundle, "models", name))
    rel.append("models/" + name)
with open(os.path.join(bundle, "models", "broken.json"), "w") as f:
    f.write('{"in_shape": [null, null, 1], "layers": [{"type": "lstm", "shape": [null, null, 12], "weights": [[1, 2')
rel += ["models/broken.json", "models/missing.json"]


STRICT = os.environ["AIDAX_STRICT_REFERENCE_SET"] == "1"
TABLE = (8, 12, 16, 20, 24, 32, 40, 64, 80)                            # variant/generate_variant_hpp.py:3-6


def loadable(path):
    """what work() will make of the file: the parsed model, or None (no answer)"""
    try:
        spec = O.load_model(path)
    except Exception:
        return None
    in_table = len(spec.layers) == 2 and spec.rnn_type in ("lstm", "gru") and spec.hidden in TABLE
    return spec if in_table or not STRICT else None


h = lv2host.Host(bundle_dir=bundle, block=512)
assert h.handle
plug = O.OraclePlugin()
controls = lambda: O.default_controls(**{lv2host.FIELD[k]: h.ctl[k].value for k in lv2host.FIELD})
RANGES = {"ANTIALIASING": (0, 100), "PREGAIN": (-12, 3), "NETBYPASS": (0, 1), "PARAM1": (0, 1), "PARAM2": (0, 1), "EQBYPASS": (0, 1),
          "EQPOS": (0, 1), "BASS": (-8, 8), "BFREQ": (75, 600), "MID": (-8, 8), "MFREQ": (150, 5000), "MIDQ": (0.2, 5), "MTYPE": (0, 1),
          "TREBLE": (-8, 8), "TFREQ": (1000, 20000), "DEPTH": (-8, 8), "PRESENCE": (-8, 8), "DCBLOCKER": (0, 1), "MASTER": (-15, 15),
          "enabled": (0, 1)}
TOGGLES = {"NETBYPASS", "EQBYPASS", "EQPOS", "MTYPE", "DCBLOCKER", "enabled"}
answers = []            # specs of the loads work() has answered, in the order of h.responses
answer_paths = []
playing = None          # path of the model work_response() swapped in last
worst, swaps, failed, last_in_size = 0.0, 0, 0, 0
for b in range(blocks):
    r = rs.rand()
    asked = None
    if r < 0.06 or b == 1:
        asked = rel[rs.randint(len(rel))] if b != 1 else rel[0]
        h.send_patch_set(os.path.join(bundle, asked))
    elif r < 0.09:
        assert h.restore(rel[rs.randint(len(rel))]) == 0               # schedules a load; does not mute (:741-755)
    elif r < 0.12:
        h.desc.activate(h.handle); plug.activate()
    if rs.rand() < 0.3:
        for k in rs.choice(sorted(RANGES), size=rs.randint(1, 4), replace=False):
            lo, hi = RANGES[k]
            v = float(rs.rand() > (0.25 if k == "enabled" else 0.5)) if k in TOGGLES else float(rs.uniform(lo, hi))
            if k == "ANTIALIASING" and rs.rand() < 0.2: v = 0.0
            h.controls(**{k: v})
    n = int(rs.choice([256, 256, 128, 512, 64, 31, 1, 0]))
    x = rs.uniform(-0.6, 0.6, size=n).astype(np.float32)
    got = h.run(x)
    if asked is not None:
        plug.set_loading(True)                                        # :576, in the run() that saw the message
    want = plug.run(controls(), x)
    if n:
        e = float(np.abs(got - want).max())
        worst = max(worst, e)
        # the bar is 1e-5 at the MODEL's output (the reference's own testModel threshold; the bundled high-gain models
        # sit at 1-3e-6 on full-scale noise); what follows the model — post EQ boosts, master gain — scales it
        lin = lambda db: 10.0 ** (max(0.0, db) / 20.0)
        g_post = lin(h.ctl["MASTER"].value)
        if h.ctl["EQPOS"].value < 0.5 and h.ctl["EQBYPASS"].value < 0.5:
            for k in ("BASS", "MID", "TREBLE", "DEPTH", "PRESENCE"): g_post *= lin(h.ctl[k].value)
        if e > 1e-5 * g_post:
            print("MISMATCH block", b, "n", n, "err", e, "asked", asked, {k: h.ctl[k].value for k in lv2host.FIELD}); sys.exit(1)
    assert h.ctl["ModelInSize"].value == last_in_size                 # what the WORKER loaded last (:828 -> :1082), shown by run() (:518)
    # the worker thread gets to its queue now or later; what it answers is known from the messages themselves
    if h.work_queue and rs.rand() < 0.6:
        for msg in h.work_queue:
            if int.from_bytes(msg[:4], "little") == 0:                # kWorkerLoad
                spec = loadable(msg[4:].split(b"\0")[0].decode())
                if spec is not None:
                    # work() reads the PARAM targets of the model that plays NOW (:822-825), not of the one that plays
                    # when the answer is delivered
                    old = plug.model.ptr.contents if plug.model is not None else None
                    answers.append((spec, old.param1Coeff.target if old else 0.0, old.param2Coeff.target if old else 0.0))
                    answer_paths.append(msg[4:].split(b"\0")[0].decode())
                    last_in_size = spec.input_size
                else: failed += 1
        h.pump_worker()
        assert len(h.responses) == len(answers)
    # ... and the host hands the answers over after some run(), not necessarily the next
    if h.responses and rs.rand() < 0.6:
        k = h.deliver_responses()
        for spec, p1, p2 in answers[:k]:
            plug.set_model(O.OracleModel(spec, p1, p2))
            swaps += 1
        playing = answer_paths[k - 1]
        del answers[:k]; del answer_paths[:k]
        # work_response() echoes the file now in use as a patch:Set on NOTIFY (:880-887); with several answers in one
        # delivery the buffer holds one event per swap, the last one naming the model that plays from here on
        notes = h.read_notify()
        assert len(notes) == k and notes[-1][0].endswith("patch#Set"), notes
        assert notes[-1][1]["http://lv2plug.in/ns/ext/patch#value"][1].rstrip(b"\0").decode() == playing
    if rs.rand() < 0.05:
        rc_, stored = h.save()                                        # :763-796: the playing model's path, abstract, as atom:Path
        assert rc_ == 0
        if playing is None: assert stored == []
        else: assert len(stored) == 1 and stored[0][1].rstrip(b"\0").decode() == os.path.relpath(playing, bundle) and stored[0][2].endswith("atom#Path")
h.close()
print("lv2 soak ok:", blocks, "blocks,", swaps, "swaps,", failed, "failed loads, worst |err| =", worst)
