"""k_lstm_q4 (aidax_q4.hip): LSTM-32 snapshot models, four streams per workgroup with the cell on
v_mfma_f32_4x4x1 and spread over the CU's four SIMDs, the chain passes on two helper waves — the A/B partner of the
3-wave pipeline at BASELINE cfg2's size (AIDAX_KERNEL=q4; measured slower, so the pool never picks it by itself:
profiles/r03_cfg2_q4_ab.txt). Kept correct all the same. Against the CPU oracle through the C ABI: the pure chain bit-exact,
anything through the network within the reference's bar; ragged stream counts and block lengths, per-stream
disable / bypass / EQ position, state carry-over, model swap, and agreement with the one-wave kernels."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O
from tests import errlog, modelgen

pytestmark = pytest.mark.gpu

ax = importlib.import_module("aidadsp-lv2_amd")
LSTM32 = dict(kind="lstm", hidden=32, input_size=1, seed=32)


def _model_file(tmp_path, name, **kw):
    j = modelgen.make_model(**kw)
    p = str(tmp_path / f"{name}.json")
    modelgen.write_model(j, p)
    return p, O.parse_model(j)


@pytest.fixture(autouse=True)
def _force_q4(monkeypatch):
    monkeypatch.setenv("AIDAX_KERNEL", "q4")


def _run(pool, x, sizes):
    out, pos = [], 0
    for n in sizes:
        out.append(pool.process(np.ascontiguousarray(x[:, pos:pos + n])))
        pos += n
    return np.concatenate(out, axis=1)


@pytest.mark.parametrize("S", [1, 3, 4, 5, 70, 1024])
def test_q4_against_the_oracle_over_stream_counts(S, tmp_path):
    path, spec = _model_file(tmp_path, "m", **LSTM32)
    sizes = [256, 256, 1, 37, 256, 100, 256]
    base = modelgen.signal(min(S, 8), sum(sizes), seed=40 + S)
    idx = np.arange(S) % base.shape[0]
    x = base[idx]
    ckw = dict(pregain_db=2.0, bass_boost_db=3.0, master_db=-1.0)
    pool = ax.Pool(S, 256)
    pool.set_model(ax.Model(path))
    pool.set_controls(ax.default_controls(**ckw))
    assert pool.kernel_name == "k_lstm_q4<32>"
    got = _run(pool, x, sizes)
    want = O.run_streams(spec, O.default_controls(**ckw), base, 4096)[idx] if False else None
    # the oracle block by block (block boundaries matter to nothing but the smoothers' targets: constant here)
    plugs = [O.OraclePlugin() for _ in range(base.shape[0])]
    for p in plugs:
        p.set_model(O.OracleModel(spec))
    co = O.default_controls(**ckw)
    pos = 0
    for n in sizes:
        for k, p in enumerate(plugs):
            w = p.run(co, base[k, pos:pos + n])
            rows = np.nonzero(idx == k)[0]
            errlog.bound(np.abs(got[rows, pos:pos + n] - w).max(), 1.5e-6, "gpu_q4:oracle")
            assert np.all(got[rows, pos:pos + n] == got[rows[0], pos:pos + n])     # identical streams, identical bits wherever they sit
        pos += n
    assert want is None
    pool.close()


def test_q4_per_stream_controls_bypass_disable_and_eq_position(tmp_path):
    """Streams of one workgroup with different lives: disabled (raw copy, nothing moves), network bypassed (pure chain:
    bit-exact), EQ in front of the network, LPF / DC blocker out of circuit, bandpass mid — every stream against its
    own oracle plugin, controls changing between blocks, incl. a pre-run (n = 0) call."""
    path, spec = _model_file(tmp_path, "m", **dict(LSTM32, in_skip=1, in_gain=-3.0, out_gain=2.0))
    S = 11
    kws = [dict(), dict(enabled=0.0), dict(net_bypass=1.0, treble_boost_db=4.0), dict(eq_position=1.0, bass_boost_db=5.0, mid_type=1.0),
           dict(dc_blocker=0.0, in_lpf_pc=0.0, eq_bypass=1.0), dict(pregain_db=6.0, master_db=-6.0, depth_boost_db=3.0), dict(net_bypass=1.0, enabled=1.0, eq_bypass=1.0)]
    flip = dict(master_db=-3.0, pregain_db=-2.0, presence_boost_db=2.0)
    sizes = [128, 0, 256, 5, 256, 64]
    x = modelgen.signal(S, sum(sizes), seed=9)
    pool = ax.Pool(S, 256)
    pool.set_model(ax.Model(path))
    assert pool.kernel_name == "k_lstm_q4<32>"
    plugs = [O.OraclePlugin() for _ in range(S)]
    for p in plugs:
        p.set_model(O.OracleModel(spec))
    pos = 0
    for b, n in enumerate(sizes):
        cur = [dict(kws[s % len(kws)], **(flip if b >= 3 and s % 2 == 0 else {})) for s in range(S)]
        for s in range(S):
            pool.set_controls(ax.default_controls(**cur[s]), stream=s)
        if b == 2:
            pool.activate(stream=4)
            plugs[4].activate()
        blk = np.ascontiguousarray(x[:, pos:pos + n])
        got = pool.process(blk)
        for s in range(S):
            want = plugs[s].run(O.default_controls(**cur[s]), blk[s])
            if cur[s].get("net_bypass") or cur[s].get("enabled") == 0.0:
                assert np.array_equal(got[s], want), (b, s)
            elif n:
                errlog.bound(np.abs(got[s] - want).max(), 4e-6, "gpu_q4:controls")
        pos += n
    pool.close()


def test_q4_agrees_with_the_one_wave_kernels_and_swaps_models(tmp_path, monkeypatch):
    """The same blocks through k_lstm_q4 and through the one-wave-per-stream pipeline: outputs and recurrent state agree
    to rounding (the dot products are summed in another order); a model swap to a GRU (another form) and back keeps
    the oracle's audio."""
    path, spec = _model_file(tmp_path, "a", **LSTM32)
    pg, sg_ = _model_file(tmp_path, "g", kind="gru", hidden=16, input_size=1, seed=7)
    S, n = 37, 256
    x = modelgen.signal(S, n * 6, seed=3)
    outs, states = {}, {}
    for form in ("q4", "pipe"):
        monkeypatch.setenv("AIDAX_KERNEL", form)
        pool = ax.Pool(S, n)
        pool.set_model(ax.Model(path))
        assert pool.kernel_name == ("k_lstm_q4<32>" if form == "q4" else "k_lstm_pipe<32>")
        outs[form] = _run(pool, x[:, :n * 3], [n] * 3)
        states[form] = pool.read_state(stream=S - 1, layer=0, hidden=32)
        pool.close()
    errlog.bound(np.abs(outs["q4"] - outs["pipe"]).max(), 1e-6, "gpu_q4:vs_pipe")
    errlog.bound(max(np.abs(states["q4"][0] - states["pipe"][0]).max(), np.abs(states["q4"][1] - states["pipe"][1]).max()), 1e-6, "gpu_q4:state_vs_pipe")
    monkeypatch.setenv("AIDAX_KERNEL", "q4")
    pool = ax.Pool(S, n)
    pool.set_model(ax.Model(path))
    assert pool.kernel_name == "k_lstm_q4<32>"
    plugs = [O.OraclePlugin() for _ in range(3)]
    for p in plugs:
        p.set_model(O.OracleModel(spec))
    co = O.default_controls()
    seq = [(path, spec), (pg, sg_), (path, spec)]
    for b in range(6):
        if b in (2, 4):
            f, sp = seq[b // 2]
            pool.set_model(ax.Model(f))
            for p in plugs:
                old = p.model.ptr.contents
                p.set_model(O.OracleModel(sp, old.param1Coeff.target, old.param2Coeff.target))
        blk = np.ascontiguousarray(x[:, b * n:(b + 1) * n])
        got = pool.process(blk)
        for k, s in enumerate((0, 17, S - 1)):
            errlog.bound(np.abs(got[s] - plugs[k].run(co, blk[s])).max(), 2e-6, "gpu_q4:swap")
    pool.close()
