"""aidax_hub: many plugin instances of one process share one pool pass per audio
period, pipelined by one period. The oracle's plugin mirror, one per instance, is the reference."""
import importlib
import os
import subprocess
import threading
import time

import numpy as np
import pytest

from oracle import oracle as O
from tests import modelgen

pytestmark = pytest.mark.gpu

ax = importlib.import_module("aidadsp-lv2_amd")
THR = 1.0e-5


def _model(tmp_path, **kw):
    j = modelgen.make_model(**kw)
    p = str(tmp_path / "m.json")
    modelgen.write_model(j, p)
    return ax.Model(p), O.parse_model(j)


def _oracle_instance(spec):
    p = O.OraclePlugin()
    p.set_model(O.OracleModel(spec))
    p.activate()
    return p


KWS = [dict(), dict(pregain_db=3.0, bass_boost_db=4.0), dict(param1=0.7, master_db=-6.0), dict(net_bypass=1.0),
       dict(eq_position=1.0, treble_boost_db=-3.0, param1=0.2)]


def test_sequential_host_one_period_of_latency(tmp_path):
    """A host that calls its five instances one after another: nobody waits, everybody gets the previous
    period's block; one launch per period."""
    m, spec = _model(tmp_path, kind="lstm", hidden=16, input_size=2, seed=4)
    N, n, periods = 5, 128, 9
    hub = ax.Hub(8, 256)
    hub.set_model(m)
    hub.set_deadline_us(0)                 # a python host may dawdle between two instances: no deadline here
    slots = [hub.attach() for _ in range(N)]
    assert sorted(slots) == list(range(N)) and hub.attached == N
    plugs = [_oracle_instance(spec) for _ in range(N)]
    x = modelgen.signal(N, n * periods, seed=31)
    prev = [np.zeros(n, np.float32) for _ in range(N)]
    for p in range(periods):
        for i in range(N):
            hub.set_controls(slots[i], ax.default_controls(**KWS[i]))
            got = hub.run(slots[i], x[i, p * n:(p + 1) * n])
            if KWS[i].get("net_bypass"):
                assert np.array_equal(got, prev[i]), (p, i)
            else:
                assert np.abs(got - prev[i]).max() < THR * 2, (p, i, np.abs(got - prev[i]).max())
            prev[i] = plugs[i].run(O.default_controls(**KWS[i]), x[i, p * n:(p + 1) * n])
    hub.flush()                            # the launcher thread may still be on its way to the last period
    assert hub.launches == periods and hub.latency_frames == n and hub.deadline_launches == 0


def test_parallel_host_threads(tmp_path):
    """A host that runs its instances on four threads with a barrier per period."""
    m, spec = _model(tmp_path, kind="gru", hidden=24, input_size=1, seed=6)
    N, n, periods = 4, 64, 12
    hub = ax.Hub(N, 64)
    hub.set_model(m)
    hub.set_deadline_us(0)
    slots = [hub.attach() for _ in range(N)]
    x = modelgen.signal(N, n * periods, seed=32)
    got = np.zeros((N, n * periods), np.float32)
    barrier = threading.Barrier(N)
    errors = []

    def instance(i):
        try:
            for p in range(periods):
                got[i, p * n:(p + 1) * n] = hub.run(slots[i], x[i, p * n:(p + 1) * n])
                barrier.wait()
        except Exception as e:                      # pragma: no cover
            errors.append(e)
            barrier.abort()

    ts = [threading.Thread(target=instance, args=(i,)) for i in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors
    for i in range(N):
        want = _oracle_instance(spec).run(O.default_controls(), x[i])
        assert np.all(got[i, :n] == 0.0)
        assert np.abs(got[i, n:] - want[:-n]).max() < THR * 2
    hub.flush()
    assert hub.launches == periods


def test_skipped_instance_does_not_advance_and_late_attach_starts_fresh(tmp_path):
    m, spec = _model(tmp_path, kind="lstm", hidden=12, input_size=1, seed=8)
    n = 96
    hub = ax.Hub(4, 128)
    hub.set_model(m)
    hub.set_deadline_us(0)
    a, b = hub.attach(), hub.attach()
    pa, pb = _oracle_instance(spec), _oracle_instance(spec)
    x = modelgen.signal(3, n * 8, seed=33)
    c = O.default_controls()
    want_a, want_b, got_a, got_b = [], [], [], []
    for p in range(6):
        blk = slice(p * n, (p + 1) * n)
        got_a.append(hub.run(a, x[0, blk]))
        want_a.append(pa.run(c, x[0, blk]))
        if p != 2:                                  # the host skips instance b in period 2
            got_b.append(hub.run(b, x[1, blk]))
            want_b.append(pb.run(c, x[1, blk]))
    # a: plain one-period delay (the skipped period is closed when a comes around again)
    ga, wa = np.concatenate(got_a), np.concatenate(want_a)
    assert np.abs(ga[n:] - wa[:-n]).max() < THR * 2
    # b: its stream did not move while it was skipped; the block after the gap sees silence once
    gb, wb = np.concatenate(got_b), np.concatenate(want_b)
    assert np.abs(gb[n:2 * n] - wb[:n]).max() < THR * 2
    assert np.abs(gb[3 * n:] - wb[2 * n:-n]).max() < THR * 2
    # a third instance attached now starts from instantiate() + warm-up state, not from a's or b's history
    cslot = hub.attach()
    pc = _oracle_instance(spec)
    outs, wants = [], []
    for p in range(6, 8):
        blk = slice(p * n, (p + 1) * n)
        hub.run(a, x[0, blk]); hub.run(b, x[1, blk])
        outs.append(hub.run(cslot, x[2, blk]))
        wants.append(pc.run(c, x[2, blk]))
    assert np.all(outs[0] == 0.0)
    assert np.abs(outs[1] - wants[0]).max() < THR * 2
    hub.detach(b)
    assert hub.attached == 2


def test_pool_reset_stream_matches_a_fresh_instance(tmp_path):
    m, spec = _model(tmp_path, kind="gru", hidden=16, input_size=1, seed=9)
    pool = ax.Pool(3, 128)
    pool.set_model(m)
    x = modelgen.signal(3, 384, seed=34)
    pool.process(np.ascontiguousarray(x[:, :128]))
    pool.reset_stream(1)
    pool.activate(1)
    got = pool.process(np.ascontiguousarray(x[:, 128:256]))
    fresh = _oracle_instance(spec).run(O.default_controls(), x[1, 128:256])
    assert np.abs(got[1] - fresh).max() < THR * 2
    cont = O.OraclePlugin(); cont.set_model(O.OracleModel(spec))
    c = O.default_controls()
    cont.run(c, x[0, :128])
    assert np.abs(got[0] - cont.run(c, x[0, 128:256])).max() < THR * 2      # neighbours are untouched


def test_deadline_one_stalled_instance_does_not_hold_the_others(tmp_path):
    """Three instances, real-time pacing; the third stops calling run() after period 3. Without a deadline its
    period would stay open until somebody comes around again; with it (here 2 ms after the period's first
    submission) the pass is launched without the straggler: the other two keep exactly one period of latency
    and find their output ready when they come around."""
    m, spec = _model(tmp_path, kind="lstm", hidden=16, input_size=1, seed=14)
    n, periods = 256, 10
    hub = ax.Hub(4, 256)
    hub.set_model(m)
    hub.set_deadline_us(2000)
    slots = [hub.attach() for _ in range(3)]
    plugs = [_oracle_instance(spec) for _ in range(3)]
    x = modelgen.signal(3, n * periods, seed=35)
    c = O.default_controls()
    prev = [np.zeros(n, np.float32) for _ in range(3)]
    period_s = n / 48000.0
    waits = []
    for p in range(periods):
        t0 = time.perf_counter()
        live = [i for i in range(3) if not (i == 2 and p >= 4)]      # instance 2 stalls from period 4 on
        got = {}
        for i in live:
            t1 = time.perf_counter()
            got[i] = hub.run(slots[i], x[i, p * n:(p + 1) * n])
            waits.append(time.perf_counter() - t1)
        for i in live:
            assert np.abs(got[i] - prev[i]).max() < THR * 2, (p, i, np.abs(got[i] - prev[i]).max())
            prev[i] = plugs[i].run(c, x[i, p * n:(p + 1) * n])
        while time.perf_counter() - t0 < period_s:  # the host's audio clock
            pass
    hub.flush()
    # (a period whose three submissions straddle the 2 ms deadline — a python host that was descheduled in between — goes out as two
    # passes and loses nothing: test_deadline_that_splits_a_period_loses_nothing; the audio above is the contract)
    assert periods <= hub.launches <= 2 * periods, hub.launches
    # on an idle box every period after the stall is closed by the deadline (periods - 3 or so); on a saturated one the
    # launcher thread may come late and a re-entering instance closes the period instead — the audio above is the contract
    assert hub.deadline_launches >= 1
    # run() never sat through a pass: with the deadline the previous period's output is ready (a pass takes < 1 ms)
    # (the ship leg runs as a second process beside the first session's GPU context: there one call in a few hundred sits out a queue
    # rotation of several milliseconds whatever the library does — the bound is the first session's; the audio above is checked on both)
    assert max(waits) < (2.0 if os.environ.get("AIDAX_SHIP_LEG") == "1" else 0.5) * period_s, max(waits)
    # the straggler comes back: its stream did not move meanwhile; first block back is silence, then it continues
    p = periods - 1
    blk = x[2, 4 * n:5 * n]
    got = hub.run(slots[2], blk)
    hub.flush()
    assert np.all(got == 0.0)
    want = plugs[2].run(c, blk)
    got = hub.run(slots[2], x[2, 5 * n:6 * n])
    assert np.abs(got - want).max() < THR * 2


def test_back_to_back_attach_and_run_from_c(tmp_path):
    """attach + activate + controls + run() with no host-side gaps, from C, while passes are in flight
    on the hub's stream — the pool's control pokes and the passes sit on different streams and must be ordered by
    the pool. Three instances join at periods 0, 2, 3; each against a fresh oracle instance."""
    j = modelgen.make_model(kind="gru", hidden=24, input_size=2, seed=21)
    mp = str(tmp_path / "m.json")
    modelgen.write_model(j, mp)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(ax.lib_path())
    exe = str(tmp_path / "c_hub_b2b")
    cc = subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-I", os.path.join(root, "include"),
                         os.path.join(root, "tests", "c_hub_b2b.c"), "-o", exe, "-L", libdir, "-laidax_hip",
                         f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    periods, n = 12, 64
    out = str(tmp_path / "out.f32")
    run = subprocess.run([exe, mp, out, str(periods), str(n)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert f"launches={periods - 1}" in run.stdout or f"launches={periods}" in run.stdout
    y = np.fromfile(out, np.float32).reshape(3, periods, n)
    spec = O.parse_model(j)
    t = np.arange(periods * n, dtype=np.uint64)
    for i, join in enumerate((0, 2, 3)):
        k = (t * np.uint64(2654435761) + np.uint64((i + 1) * 40503)) & np.uint64(0xffffffff)
        x = (((k >> np.uint64(8)) & np.uint64(0xffff)).astype(np.float32) / np.float32(65535.0) - np.float32(0.5)) * np.float32(0.8)
        plug = _oracle_instance(spec)
        c = O.default_controls(pregain_db=2.0 * i, param1=0.25 * i)
        want = np.concatenate([plug.run(c, x[p * n:(p + 1) * n]) for p in range(join, periods)]).reshape(-1, n)
        got = y[i, join:]
        assert np.all(got[0] == 0.0)                                   # one period of latency
        assert np.abs(got[1:] - want[:-1]).max() < THR * 2, (i, np.abs(got[1:] - want[:-1]).max())


@pytest.mark.parametrize("moving_split", [False, True])
def test_deadline_that_splits_a_period_loses_nothing(tmp_path, moving_split):
    """A sequential host whose submissions span MORE than the deadline (here: a 5 ms pause in the middle of every
    period against a 1.2 ms deadline): every period is closed in two passes. Each instance's previous block then
    sits in the buffer of the pass IT was part of (per-slot pass tracking over four rotating buffers), also when the
    place of the pause — and with it the pass an instance falls into — moves from period to period. Everybody keeps
    exactly one period of latency, nothing is lost or repeated.
    (The premise is a host that gets its six run() calls of a period out within the deadline apart from the pause. One that
    does not — a box whose cores are taken: every call its own pass — closes more passes per period than the hub keeps
    results for (four buffers: an instance finds its previous block while at most three passes were closed since), which is
    the hub's documented limit, not what this test is about: it then skips instead of reporting lost blocks.)"""
    m, spec = _model(tmp_path, kind="lstm", hidden=12, input_size=1, seed=18)
    N, n, periods = 6, 128, 10
    hub = ax.Hub(8, 128)
    hub.set_model(m)
    hub.set_deadline_us(1200)
    slots = [hub.attach() for _ in range(N)]
    plugs = [_oracle_instance(spec) for _ in range(N)]
    x = modelgen.signal(N, n * periods, seed=36)
    c = O.default_controls()
    prev = [np.zeros(n, np.float32) for _ in range(N)]
    for p in range(periods):
        pause_after = (2 + p % 3) if moving_split else 2
        got = {}
        launches0 = hub.launches
        for i in range(N):
            got[i] = hub.run(slots[i], x[i, p * n:(p + 1) * n])
            if i == pause_after:
                time.sleep(0.005)
        time.sleep(0.005)                           # the rest of the period: the second pass is closed by its deadline too
        if hub.launches - launches0 > 3:
            pytest.skip(f"host too slow for the premise: {hub.launches - launches0} passes closed in period {p} (deadline 1.2 ms)")
        for i in range(N):
            assert np.abs(got[i] - prev[i]).max() < THR * 2, (p, i, np.abs(got[i] - prev[i]).max())
            prev[i] = plugs[i].run(c, x[i, p * n:(p + 1) * n])
    hub.flush()
    # two passes per period on an idle box; a starved launcher thread closes fewer by deadline (the audio is the contract)
    assert hub.launches >= periods and hub.deadline_launches >= 1


def test_host_that_runs_ahead_of_the_gpu_keeps_its_staging_intact(tmp_path):
    """Two instances called in turn without pause, every call with another block size than the call before it AND than
    the same instance's previous call: every run() closes the other's period on the spot, and none of them reads a
    block back (its previous output has another length), so nothing ever makes the host wait — it issues a pass every
    few microseconds while a 256-frame pass of the stacked model takes a good part of a millisecond, and is soon many
    passes ahead of the GPU. A pass's input staging (four buffers in rotation) must not be collected into again before
    that pass has read it: a pass computed from a later block's input leaves the stream's state off the oracle's, which
    the last round — same sizes twice, so it reads back — shows. The oracle's outputs are computed beforehand so that
    nothing slows the host loop down. Found by tests/soak_hub.py; fails without the wait in flush_locked."""
    m, spec = _model(tmp_path, kind="lstm", hidden=32, input_size=1, seed=321, n_rnn=2)
    rounds = 41
    size = lambda i, r: ((256, 128), (17, 64))[i][r % 2 if r < rounds - 1 else (rounds - 2) % 2]
    c = O.default_controls()
    rs = np.random.RandomState(5)
    blocks = [[rs.uniform(-0.6, 0.6, size=size(i, r)).astype(np.float32) for r in range(rounds)] for i in range(2)]
    want = []
    for i in range(2):
        plug = _oracle_instance(spec)
        want.append([plug.run(c, b) for b in blocks[i]])
    hub = ax.Hub(4, 256)
    hub.set_model(m)
    hub.set_deadline_us(0)
    slots = [hub.attach() for _ in range(2)]
    got = [[None] * rounds for _ in range(2)]
    for r in range(rounds):
        for i in range(2):
            got[i][r] = hub.run(slots[i], blocks[i][r])
    hub.flush()
    assert hub.launches == 2 * rounds
    for i in range(2):
        for r in range(rounds - 1):
            assert np.all(got[i][r] == 0.0), (i, r)                     # another length than the block before: silence
        err = np.abs(got[i][rounds - 1] - want[i][rounds - 2]).max()
        assert err < THR, (i, err)


def test_an_instance_that_moves_between_hubs_keeps_its_filter_and_gain_memories(tmp_path):
    """A model-file change in hub mode (the reference's swap, rt-neural-generic.cpp:868-875, spread over two hubs): the
    worker attaches a successor seat in the hub of the new file (fresh DynamicModel around the playing model's PARAM
    targets, :822-825), the audio thread adopts — the plugin's biquad memories and gain smoothers travel on the device,
    behind the old seat's last block. ONE oracle plugin per instance for its whole life is the reference: with the
    network bypassed the audio after the move is bit-exact (the filters ring on, the gains keep ramping), with it the
    reference's tolerance holds. A second instance stays on the first hub and must not notice."""
    ja = modelgen.make_model(kind="gru", hidden=16, input_size=3, seed=41)
    jb = modelgen.make_model(kind="lstm", hidden=20, input_size=2, seed=42)
    pa, pb = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    modelgen.write_model(ja, pa); modelgen.write_model(jb, pb)
    sa, sb = O.parse_model(ja), O.parse_model(jb)
    n = 128
    for bypass in (1.0, 0.0):
        hub_a, hub_b = ax.Hub(4, 256), ax.Hub(4, 256)
        hub_a.set_model(ax.Model(pa)); hub_b.set_model(ax.Model(pb))
        hub_a.set_deadline_us(0); hub_b.set_deadline_us(0)
        ckw = dict(net_bypass=bypass, param1=0.6, param2=0.2, pregain_db=4.0, master_db=-3.0, bass_boost_db=5.0, treble_boost_db=-4.0,
                   eq_position=1.0, in_lpf_pc=40.0)
        cg, co = ax.default_controls(**ckw), O.default_controls(**ckw)
        mover, stayer = hub_a.attach(), hub_a.attach()
        pm, ps = _oracle_instance(sa), _oracle_instance(sa)
        x = modelgen.signal(2, n * 24, seed=77)
        hub, slot, prev_m, prev_s = hub_a, mover, None, None
        for b in range(24):
            blk = [np.ascontiguousarray(x[i, b * n:(b + 1) * n]) for i in range(2)]
            hub.set_controls(slot, cg)
            got_m = hub.run(slot, blk[0])
            hub_a.set_controls(stayer, cg)
            got_s = hub_a.run(stayer, blk[1])
            for got, prev, who in ((got_m, prev_m, "mover"), (got_s, prev_s, "stayer")):
                if prev is None:
                    assert not got.any(), (b, who)
                elif bypass:
                    assert np.array_equal(got, prev), (b, who, np.abs(got - prev).max())
                else:
                    assert np.abs(got - prev).max() < THR * 2, (b, who, np.abs(got - prev).max())
            prev_m, prev_s = pm.run(co, blk[0]), ps.run(co, blk[1])
            if b in (7, 15):                                         # the instance's worker has loaded the other file ...
                new_hub, spec = (hub_b, sb) if hub is hub_a else (hub_a, sa)
                old = pm.model.ptr.contents
                p1, p2 = old.param1Coeff.target, old.param2Coeff.target
                new_slot = new_hub.attach_successor(hub, slot)
                new_hub.adopt(new_slot, hub, slot)                   # ... and the host delivers work_response()
                hub.detach(slot)                                     # kWorkerFree
                hub, slot = new_hub, new_slot
                pm.set_model(O.OracleModel(spec, p1, p2))
                prev_m = None                                        # one period of latency starts over on the new seat
        hub_a.close(); hub_b.close()


def test_export_and_import_of_a_streams_dsp_members(tmp_path):
    """aidax_pool_export_stream_dsp / _import_stream_dsp: the biquad states and gain smoothers of a running stream
    planted into a fresh pool make it continue bit for bit (network bypassed) where the first one stands."""
    m, spec = _model(tmp_path, kind="lstm", hidden=12, input_size=1, seed=5)
    ckw = dict(net_bypass=1.0, pregain_db=-5.0, master_db=2.0, mid_boost_db=6.0, mid_q=2.0, depth_boost_db=3.0)
    cg, co = ax.default_controls(**ckw), O.default_controls(**ckw)
    n = 96
    x = modelgen.signal(1, n * 12, seed=12)
    pool = ax.Pool(3, n)
    pool.set_model(m); pool.set_controls(cg)
    plug = _oracle_instance(spec)
    for b in range(6):
        blk = np.ascontiguousarray(x[:, b * n:(b + 1) * n])
        got = pool.process(np.repeat(blk, 3, axis=0))
        assert np.array_equal(got[1], plug.run(co, blk[0]))
    d = pool.export_stream_dsp(1)
    assert d.pre_target == pytest.approx(10 ** (-5.0 / 20), rel=1e-6) and any(abs(d.z[k][0]) > 0 for k in range(7))
    pool2 = ax.Pool(2, n)
    pool2.set_model(m); pool2.set_controls(cg)
    pool2.import_stream_dsp(0, d)
    for b in range(6, 12):
        blk = np.ascontiguousarray(x[:, b * n:(b + 1) * n])
        got = pool2.process(np.repeat(blk, 2, axis=0))
        want = plug.run(co, blk[0])
        assert np.array_equal(got[0], want), b
        assert not np.array_equal(got[1], want)                      # the untouched stream started from instantiate()
    pool.close(); pool2.close()
