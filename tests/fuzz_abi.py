# C-ABI fuzz: random sequences of entry-point calls with arguments drawn from valid AND cleanly-invalid values (null
# handles and buffers, streams / slots / sizes out of range, wrong modes, detached slots, models the pool cannot host).
# Every call must return AIDAX_OK or a negative code with a message — never crash, hang or leave the pool unusable:
# after each burst a known block goes through and must still match the oracle.
# usage: python tests/fuzz_abi.py [bursts]
import ctypes as C
import importlib, os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
B = importlib.import_module("aidadsp-lv2_amd.binding")
L = B.lib()
fp = C.POINTER(C.c_float)

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "11")))
d = tempfile.mkdtemp()
specs = [dict(kind="lstm", hidden=16, input_size=1, seed=1), dict(kind="gru", hidden=24, input_size=3, seed=2),
         dict(kind="lstm", hidden=32, input_size=1, seed=3, n_rnn=2), dict(kind="conv", hidden=16, input_size=1, seed=4),
         dict(kind="lstm", hidden=128, input_size=1, seed=5)]
models = []
for i, kw in enumerate(specs):
    j = modelgen.make_model(**kw)
    models.append((ax.Model(modelgen.write_model(j, os.path.join(d, f"m{i}.json"))), O.parse_model(j)))
calls = fails = 0


def rc(x):
    global calls, fails
    calls += 1
    if x != 0:
        fails += 1
        assert x < 0, x
        assert L.aidax_last_error(), "a failure without a message"
    return x


def probe(pool, S, MAXF, spec, sr):
    """the pool must still be a working pool: a fresh stream state and a known block against the oracle"""
    n = int(min(MAXF, 64))
    rc(L.aidax_pool_set_model(pool, models[0][0].h, 0))
    assert rc(L.aidax_pool_reset_stream(pool, 0, 0)) == 0          # stream 0 as a new instance (filter memories, ramps)
    c = ax.default_controls(pregain_db=1.0)
    rc(L.aidax_pool_set_controls(pool, -1, C.byref(c)))
    rc(L.aidax_pool_activate(pool, -1))
    x = (rs.uniform(-0.5, 0.5, size=(S, n))).astype(np.float32)
    y = np.empty_like(x)
    assert L.aidax_pool_process(pool, x.ctypes.data_as(fp), y.ctypes.data_as(fp), n) == 0, L.aidax_last_error()
    pl = O.OraclePlugin(sr); pl.set_model(O.OracleModel(models[0][1])); pl.activate()
    want = pl.run(O.default_controls(pregain_db=1.0), x[0])
    assert np.abs(y[0] - want).max() < 5e-6, np.abs(y[0] - want).max()


for b in range(bursts):
    S = int(rs.choice([1, 3, 9, 70]))
    MAXF = int(rs.choice([1, 64, 256, 2048]))
    pool = C.c_void_p()
    # cleanly-invalid creations first
    for bad in ((0, 256, 48000.0, 0), (4, 0, 48000.0, 0), (4, 1 << 20, 48000.0, 0), (4, 256, 0.0, 0), (4, 256, 48000.0, 99), (4, 256, 48000.0, -1)):
        tmp = C.c_void_p()
        assert rc(L.aidax_pool_create(bad[0], bad[1], bad[2], bad[3], C.byref(tmp))) < 0 and not tmp.value
    sr = float(rs.choice([44100.0, 48000.0, 96000.0]))
    assert rc(L.aidax_pool_create(S, MAXF, sr, 0, C.byref(pool))) == 0
    staged = []
    xbuf = np.zeros((S, MAXF), np.float32)
    for _ in range(int(rs.randint(10, 60))):
        op = rs.randint(14)
        mi = rs.randint(len(models))
        stream = int(rs.choice([-1, 0, S - 1, S, S + 5, -2, 1 << 30]))
        if op == 0:
            rc(L.aidax_pool_set_model(pool, models[mi][0].h if rs.rand() > 0.1 else None, int(rs.choice([0, 1, 7]))))
        elif op == 1:
            sg = C.c_void_p()
            if rc(L.aidax_pool_prepare_model(pool, models[mi][0].h if rs.rand() > 0.1 else None, int(rs.choice([0, 1, 7])), C.byref(sg))) == 0:
                staged.append(sg)
        elif op == 2 and staged:
            sg = staged.pop(rs.randint(len(staged)))
            rc(L.aidax_pool_commit_model(pool, sg))
            L.aidax_staged_free(sg)
        elif op == 3:
            rc(L.aidax_pool_commit_model(pool, None)); L.aidax_staged_free(None)
        elif op == 4:
            c = ax.default_controls(param1=float(rs.rand()), master_db=float(rs.uniform(-20, 20)), mid_q=float(rs.choice([0.0, 0.2, 5.0, 50.0])),
                                    bass_freq=float(rs.choice([0.0, 75.0, 30000.0])), in_lpf_pc=float(rs.choice([0.0, 66.2, 100.0, 250.0, -5.0])))
            rc(L.aidax_pool_set_controls(pool, stream if stream < (1 << 20) else -1, C.byref(c) if rs.rand() > 0.05 else None))
        elif op == 5:
            rc(L.aidax_pool_activate(pool, stream if abs(stream) < (1 << 20) else S))
        elif op == 6:
            rc(L.aidax_pool_set_loading(pool, stream if abs(stream) < (1 << 20) else S, int(rs.randint(2))))
        elif op == 7:
            rc(L.aidax_pool_reset_stream(pool, int(rs.choice([0, S - 1, S, 1 << 30])), int(rs.choice([0, 1, 5]))))
        elif op == 8:
            n = int(rs.choice([0, 1, MAXF, MAXF + 1, 17 if MAXF >= 17 else 1]))
            x = np.ascontiguousarray(xbuf[:, :max(n, 1)]) if n <= MAXF else np.zeros((S, n), np.float32)
            y = np.empty_like(x)
            null_in = rs.rand() < 0.05
            rc(L.aidax_pool_process(pool, None if null_in else x.ctypes.data_as(fp), y.ctypes.data_as(fp), n))
        elif op == 9:
            h = np.zeros(256, np.float32); cc = np.zeros(256, np.float32)
            r = L.aidax_pool_read_state(pool, int(rs.choice([0, S - 1, S])), int(rs.choice([0, 1, 5, -1])), h.ctypes.data_as(fp), cc.ctypes.data_as(fp), int(rs.choice([0, 4, 256])))
            calls += 1
            if r < 0: fails += 1
        elif op == 10:
            rc(L.aidax_pool_sync(pool)); assert L.aidax_pool_kernel_name(pool) is not None
        elif op == 11:
            rc(L.aidax_pool_process(None, None, None, 0)); rc(L.aidax_pool_set_model(None, None, 0)); rc(L.aidax_pool_sync(None))
        elif op == 12:
            info = B.ModelInfo() if hasattr(B, "ModelInfo") else None
            rc(L.aidax_model_info(None, None))
        else:
            tmp = C.c_void_p()
            assert rc(L.aidax_model_load(os.path.join(d, "nope.json").encode(), C.byref(tmp))) < 0 and not tmp.value
            assert rc(L.aidax_model_load_memory(b"{\"in_shape\": 3", 14, b"junk", C.byref(tmp))) < 0 and not tmp.value
    for sg in staged:
        L.aidax_staged_free(sg)
    if MAXF >= 1:
        probe(pool, S, MAXF, models[0][1], sr)
    L.aidax_pool_destroy(pool)
    # ---- the hub: slots that are not attached, blocks that are too long, null buffers, null handles
    hub = C.c_void_p()
    assert rc(L.aidax_hub_create(0, 64, 48000.0, 0, C.byref(hub))) < 0 and not hub.value
    assert rc(L.aidax_hub_create(4, 64, 48000.0, 0, C.byref(hub))) == 0
    rc(L.aidax_hub_set_deadline_us(hub, 0))
    rc(L.aidax_hub_set_model(hub, models[rs.randint(4)][0].h, 0))
    attached = []
    for _ in range(int(rs.randint(10, 40))):
        op = rs.randint(8)
        slot = int(rs.choice(attached + [-1, 3, 4, 99]))
        if op == 0:
            sl = C.c_int32(-1)
            if rc(L.aidax_hub_attach(hub, C.byref(sl))) == 0: attached.append(sl.value)
        elif op == 1:
            if rc(L.aidax_hub_detach(hub, slot)) == 0: attached.remove(slot)
        elif op == 2:
            n = int(rs.choice([0, 1, 64, 65]))
            x = np.zeros(max(n, 1), np.float32); y = np.empty_like(x)
            rc(L.aidax_hub_run(hub, slot, None if rs.rand() < 0.05 else x.ctypes.data_as(fp), y.ctypes.data_as(fp), n))
        elif op == 3:
            c = ax.default_controls(param1=float(rs.rand()))
            rc(L.aidax_hub_set_controls(hub, slot, C.byref(c) if rs.rand() > 0.05 else None))
        elif op == 4:
            rc(L.aidax_hub_activate(hub, slot)); rc(L.aidax_hub_set_loading(hub, slot, int(rs.randint(2))))
        elif op == 5:
            rc(L.aidax_hub_flush(hub)); L.aidax_hub_launches(hub); L.aidax_hub_latency_frames(hub); L.aidax_hub_attached(hub)
        elif op == 6:
            rc(L.aidax_hub_set_model(hub, models[rs.randint(len(models))][0].h if rs.rand() > 0.1 else None, int(rs.choice([0, 1, 7]))))
        else:
            rc(L.aidax_hub_run(None, 0, None, None, 0)); rc(L.aidax_hub_flush(None)); rc(L.aidax_hub_attach(None, None)); L.aidax_hub_launches(None)
    L.aidax_hub_destroy(hub)
    L.aidax_hub_destroy(None)
print("abi fuzz ok:", bursts, "bursts,", calls, "calls,", fails, "clean failures")
