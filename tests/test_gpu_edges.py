"""GPU edge cases of the C-ABI boundary: in-place buffers, host sample rates other than
48 kHz, model sample rate from json, argument errors as codes, the largest block, many
small pools, and the device-resident entry point on a caller stream with in-place I/O."""
import ctypes as C
import importlib
import json

import numpy as np
import pytest

from oracle import oracle as O
from tests import errlog, modelgen

pytestmark = pytest.mark.gpu
ax = importlib.import_module("aidadsp-lv2_amd")
THR = 1.0e-5
_fp = C.POINTER(C.c_float)


def _model(tmp_path, name, **kw):
    j = modelgen.make_model(**kw)
    p = str(tmp_path / f"{name}.json")
    modelgen.write_model(j, p)
    return p, O.parse_model(j)


def test_in_place_process_matches_out_of_place(tmp_path):
    path, spec = _model(tmp_path, "ip", kind="lstm", hidden=16, input_size=1, seed=3)
    x = modelgen.signal(5, 512, seed=2)
    a, b = ax.Pool(5, 256), ax.Pool(5, 256)
    m = ax.Model(path)
    a.set_model(m)
    b.set_model(m)
    L = ax.lib()
    for blk in range(2):
        chunk = np.ascontiguousarray(x[:, blk * 256:(blk + 1) * 256])
        want = a.process(chunk)
        buf = chunk.copy()
        assert L.aidax_pool_process(b.h, buf.ctypes.data_as(_fp), buf.ctypes.data_as(_fp), 256) == 0
        assert np.array_equal(buf, want)
    # disabled + in place: nothing to copy, nothing changes
    b.set_controls(ax.default_controls(enabled=0.0))
    buf = np.ascontiguousarray(x[:, :256]).copy()
    assert L.aidax_pool_process(b.h, buf.ctypes.data_as(_fp), buf.ctypes.data_as(_fp), 256) == 0
    assert np.array_equal(buf, x[:, :256])


@pytest.mark.parametrize("host_sr", [44100.0, 96000.0])
def test_host_samplerate_drives_filters_and_gain_ramps(host_sr, tmp_path):
    """Biquad Fc/fs and the gain smoothers use the HOST rate (:283-314), the PARAM smoothers the
    MODEL rate (:1053-1060); a json `samplerate` NUMBER overrides the 48 kHz default (:1005-1013)."""
    j = modelgen.make_model("gru", 12, 2, seed=9, samplerate=44100)
    path = str(tmp_path / "sr.json")
    modelgen.write_model(j, path)
    spec = O.parse_model(j)
    assert spec.samplerate == 44100.0
    x = modelgen.signal(3, 1024, seed=6, fs=host_sr)
    kw = dict(bass_boost_db=5.0, treble_boost_db=-4.0, pregain_db=4.0, master_db=-3.0, param1=0.8)
    pool = ax.Pool(3, 256, samplerate=host_sr)
    pool.set_model(ax.Model(path))
    pool.set_controls(ax.default_controls(**kw))
    got = np.concatenate([pool.process(np.ascontiguousarray(x[:, b:b + 256])) for b in range(0, 1024, 256)], axis=1)
    want = O.run_streams(spec, O.default_controls(**kw), x, 256, samplerate=host_sr)
    errlog.bound(np.abs(got - want).max(), 5e-7, "gpu_edges:64")


def test_argument_errors_are_codes(tmp_path):
    L = ax.lib()
    h = C.c_void_p()
    assert L.aidax_pool_create(0, 256, C.c_double(48000.0), 0, C.byref(h)) == -1
    assert L.aidax_pool_create(4, 0, C.c_double(48000.0), 0, C.byref(h)) == -1
    assert L.aidax_pool_create(4, 1 << 20, C.c_double(48000.0), 0, C.byref(h)) == -1
    assert L.aidax_pool_create(4, 256, C.c_double(0.0), 0, C.byref(h)) == -1
    assert L.aidax_pool_create(4, 256, C.c_double(48000.0), 99, C.byref(h)) == -1
    pool = ax.Pool(4, 64)
    x = np.zeros((4, 128), np.float32)
    assert L.aidax_pool_process(pool.h, x.ctypes.data_as(_fp), x.ctypes.data_as(_fp), 128) == -1      # > max_frames
    assert L.aidax_pool_process(pool.h, None, None, 64) == -1
    assert L.aidax_pool_process(pool.h, None, None, 0) == 0                                           # pre-run needs no buffers
    c = ax.default_controls()
    assert L.aidax_pool_set_controls(pool.h, 4, C.byref(c)) == -1 and L.aidax_pool_set_controls(pool.h, -2, C.byref(c)) == -1
    assert L.aidax_pool_activate(pool.h, 7) == -1 and L.aidax_pool_set_loading(pool.h, 9, 1) == -1
    assert L.aidax_pool_set_model(pool.h, None, 5) == -1
    assert b"" != L.aidax_last_error()
    # a conv model with params is an architecture error at load time
    rc = C.c_void_p()
    jj = modelgen.make_model("conv", 16, 2, seed=1, conv_layers=2)
    txt = json.dumps(jj).encode()
    assert L.aidax_model_load_memory(txt, len(txt), b"x", C.byref(rc)) == -4
    # unloading a model mutes the pool again (loading = true)
    path, _ = _model(tmp_path, "u", kind="lstm", hidden=8, input_size=1, seed=1)
    pool.set_model(ax.Model(path))
    sig = modelgen.signal(4, 64, seed=1)
    assert np.abs(pool.process(sig)).max() > 0
    pool.set_model(None)
    for _ in range(200):
        out = pool.process(sig)
    errlog.bound(np.abs(out).max(), 1e-6, "gpu_edges:98")


def test_largest_block_and_block_size_invariance(tmp_path):
    """8192-frame blocks (the pool maximum) equal 32 x 256-frame blocks: bit-exact for the pure
    chain, within tolerance through the NN (all three forms see n > one pipeline ring)."""
    path, spec = _model(tmp_path, "big", kind="lstm", hidden=24, input_size=1, seed=4)
    x = modelgen.signal(2, 8192, seed=12)
    m = ax.Model(path)
    a, b = ax.Pool(2, 8192), ax.Pool(2, 256)
    a.set_model(m)
    b.set_model(m)
    big = a.process(x)
    small = np.concatenate([b.process(np.ascontiguousarray(x[:, i:i + 256])) for i in range(0, 8192, 256)], axis=1)
    errlog.bound(np.abs(big - small).max(), 1e-7, "gpu_edges:112")
    want = O.run_streams(spec, O.default_controls(), x, 8192)
    errlog.bound(np.abs(big - want).max(), 1e-6, "gpu_edges:114")
    c, d = ax.Pool(2, 8192), ax.Pool(2, 100)
    for p in (c, d):
        p.set_loading(False)
    assert np.array_equal(c.process(x), np.concatenate([d.process(np.ascontiguousarray(x[:, i:i + 100])) for i in range(0, 8192, 100)], axis=1))


def test_many_pools_coexist_and_stay_independent(tmp_path):
    paths = [_model(tmp_path, f"m{i}", kind=("lstm", "gru")[i % 2], hidden=(8, 12, 16, 20)[i % 4], input_size=1 + i % 3, seed=40 + i)
             for i in range(8)]
    x = modelgen.signal(3, 384, seed=77)
    pools = []
    for p, _ in paths:
        pool = ax.Pool(3, 128)
        pool.set_model(ax.Model(p))
        pools.append(pool)
    outs = [np.concatenate([pool.process(np.ascontiguousarray(x[:, b:b + 128])) for b in range(0, 384, 128)], axis=1) for pool in pools]
    for (p, spec), got in zip(paths, outs):
        want = O.run_streams(spec, O.default_controls(), x, 128)
        errlog.bound(np.abs(got - want).max(), 5e-7, "gpu_edges:133")


def test_device_entry_point_in_place_on_caller_stream(tmp_path):
    import torch
    path, spec = _model(tmp_path, "dev2", kind="gru", hidden=32, input_size=1, seed=8)
    x = modelgen.signal(130, 256, seed=5)
    ref = ax.Pool(130, 256)
    ref.set_model(ax.Model(path))
    want = ref.process(x)
    pool = ax.Pool(130, 256)
    pool.set_model(ax.Model(path))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = torch.from_numpy(x).cuda()
        pool.process_device(d.data_ptr(), d.data_ptr(), 256, s.cuda_stream)
    s.synchronize()
    assert np.array_equal(d.cpu().numpy(), want)


def test_extreme_levels_denormal_silence_and_full_scale(tmp_path):
    """Signal levels the synthetic bench never visits: fp32-denormal input (the chain without a model must
    stay bit-exact: fp32 denormals are honoured on the device like on the reference's CPU), a second of digital
    silence after a loud passage (states decay, nothing drifts), and a full-scale square wave through the NN."""
    import importlib
    ax = importlib.import_module("aidadsp-lv2_amd")
    from oracle import oracle as O
    from tests import modelgen
    n, block = 4096, 256
    rs = np.random.RandomState(5)
    tiny = (rs.uniform(-1, 1, n) * 1e-39).astype(np.float32)
    assert np.any((np.abs(tiny) > 0) & (np.abs(tiny) < 1.1754944e-38))
    loud_then_silent = np.concatenate([modelgen.signal(1, n // 2, seed=1)[0] * 1.9, np.zeros(n // 2, np.float32)])
    square = np.where((np.arange(n) // 37) % 2 == 0, 1.0, -1.0).astype(np.float32)
    x = np.stack([tiny, loud_then_silent, square])
    # (1) chain only
    pool = ax.Pool(3, block)
    pool.set_loading(False)
    cg = ax.default_controls(bass_boost_db=6.0, treble_boost_db=-4.0, pregain_db=12.0, master_db=-15.0)
    co = O.default_controls(bass_boost_db=6.0, treble_boost_db=-4.0, pregain_db=12.0, master_db=-15.0)
    pool.set_controls(cg)
    got = np.concatenate([pool.process(np.ascontiguousarray(x[:, b:b + block])) for b in range(0, n, block)], axis=1)
    for s in range(3):
        p = O.OraclePlugin()
        p.set_loading(False)
        want = np.concatenate([p.run(co, x[s, b:b + block]) for b in range(0, n, block)])
        assert np.array_equal(got[s], want), s
    # (2) through an LSTM-32 and a GRU-16
    for kind, hidden in (("lstm", 32), ("gru", 16)):
        j = modelgen.make_model(kind, hidden, 1, seed=77)
        path = modelgen.write_model(j, str(tmp_path / f"{kind}.json"))
        spec = O.parse_model(j)
        pool = ax.Pool(3, block)
        pool.set_model(ax.Model(path))
        got = np.concatenate([pool.process(np.ascontiguousarray(x[:, b:b + block])) for b in range(0, n, block)], axis=1)
        want = O.run_streams(spec, O.default_controls(), x, block)
        assert np.isfinite(got).all()
        errlog.bound(np.abs(got - want).max(), 1.5e-6, "gpu_edges:190")


@pytest.mark.parametrize("kw,max_frames,kernel", [
    (dict(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=5), 512, "k_conv_ms"),           # blocks > 256: time slices of the one-launch form
    (dict(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=5), 512, "k_conv_mfma"),         # ... the fp32 partner (AIDAX_CONV_MS=0)
    (dict(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=5), 512, "k_conv"),              # ... and the VALU conv with the whole block in LDS (AIDAX_KERNEL=valu)
    (dict(kind="lstm", hidden=48, input_size=2, seed=4, n_rnn=2), 2048, "k_chain+k_mfma_ls"),             # longest block of the packed chains
    (dict(kind="lstm", hidden=20, input_size=1, seed=5, n_rnn=2), 256, "k_mfma_ls"),                      # 20 units run zero-padded to 32 (one launch: blocks of one staging chunk)
])
def test_extension_fallback_kernels_for_long_blocks(kw, max_frames, kernel, tmp_path, monkeypatch):
    """Conv pools with blocks above 256 frames send them through the matrix-core kernel in time slices (the VALU
    kernel takes them whole); stacked models take blocks up to 2048 frames on k_mfma (eight staging chunks per launch)
    and run zero-padded when their width is not a multiple of 16."""
    import importlib
    if kernel == "k_conv":
        monkeypatch.setenv("AIDAX_KERNEL", "valu")
    if kernel == "k_conv_mfma":
        monkeypatch.setenv("AIDAX_CONV_MS", "0")
    ax = importlib.import_module("aidadsp-lv2_amd")
    j = modelgen.make_model(**kw)
    path = modelgen.write_model(j, str(tmp_path / "m.json"))
    spec = O.parse_model(j)
    S = 5
    pool = ax.Pool(S, max_frames)
    pool.set_model(ax.Model(path))
    assert pool.kernel_name == kernel
    sizes = [max_frames, 100, max_frames - 1] if max_frames <= 512 else [700, 2048, 33]
    x = modelgen.signal(S, sum(sizes), seed=12)
    got = np.empty_like(x)
    want = np.empty_like(x)
    plugs = []
    for s in range(S):
        p = O.OraclePlugin(); p.set_model(O.OracleModel(spec)); plugs.append(p)
    pos = 0
    for n in sizes:
        got[:, pos:pos + n] = pool.process(np.ascontiguousarray(x[:, pos:pos + n]))
        for s in range(S):
            want[s, pos:pos + n] = plugs[s].run(O.default_controls(), x[s, pos:pos + n])
        pos += n
    errlog.bound(np.abs(got - want).max(), 1e-6, "gpu_edges:224")


def test_muted_tail_runs_through_the_denormals_bit_for_bit():
    """A plugin that is muted (loading: master target 0) lets its master gain decay exponentially — through the fp32
    denormal range, which the CPU reference (no -ffast-math, no flush-to-zero) keeps. 150 000 samples: a quarter of a
    million denormal output samples, every block bit-identical to the oracle."""
    import importlib
    ax = importlib.import_module("aidadsp-lv2_amd")
    S, n, blocks = 3, 256, 600
    pool = ax.Pool(S, n)
    pool.set_controls(ax.default_controls(master_db=6.0))
    pool.activate()
    plugs = [O.OraclePlugin() for _ in range(S)]
    for p in plugs:
        p.activate()
    x = modelgen.signal(S, n * blocks, seed=3)
    tiny = 0
    for b in range(blocks):
        blk = np.ascontiguousarray(x[:, b * n:(b + 1) * n])
        if b == 2:                                    # the master ramp comes up ...
            for p in plugs:
                p.set_loading(False)
            pool.set_loading(False)
        if b == 40:                                   # ... and decays towards zero for the rest of the run
            for p in plugs:
                p.set_loading(True)
            pool.set_loading(True)
        got = pool.process(blk)
        want = np.stack([plugs[s].run(O.default_controls(master_db=6.0), blk[s]) for s in range(S)])
        tiny += int(((np.abs(want) > 0) & (np.abs(want) < 1.2e-38)).sum())
        assert np.array_equal(got, want), b
    assert tiny > 100000


def test_pipelined_host_buffer_path_is_the_blocking_path_bit_for_bit(tmp_path):
    """aidax_pool_submit / aidax_pool_collect with a block kept in flight (the upload of block k+1 and the download of
    k-1 run under the pass of k) against aidax_pool_process on an identical pool: same bits, block for block, with
    controls that move between blocks and block lengths that change; the calling rules are enforced."""
    path = str(tmp_path / "p.json")
    modelgen.write_model(modelgen.make_model(kind="gru", hidden=24, input_size=2, seed=19), path)
    m = ax.Model(path)
    S, sizes = 70, [128, 128, 64, 256, 256, 1, 200, 128, 128, 128]
    x = modelgen.signal(S, sum(sizes), seed=15)
    a, b = ax.Pool(S, 256), ax.Pool(S, 256)
    a.set_model(m); b.set_model(m)
    L = ax.lib()
    out = np.empty((S, 256), np.float32)
    assert L.aidax_pool_collect(b.h, out.ctypes.data_as(C.POINTER(C.c_float)), 128) == -6        # AIDAX_ERR_STATE: nothing submitted
    blocks, pos = [], 0
    for n in sizes:
        blocks.append(np.ascontiguousarray(x[:, pos:pos + n])); pos += n
    ctl = lambda k: ax.default_controls(param1=0.1 * k, pregain_db=float(k % 3), bass_boost_db=2.0 * (k % 2))
    want = []
    for k, blk in enumerate(blocks):
        a.set_controls(ctl(k))
        want.append(a.process(blk))
    b.set_controls(ctl(0))
    b.submit(blocks[0])
    for k in range(1, len(blocks)):
        b.set_controls(ctl(k))                                   # applies to the block submitted next, not to the one in flight
        b.submit(blocks[k])
        got = b.collect(sizes[k - 1])
        assert np.array_equal(got, want[k - 1]), k
    assert np.array_equal(b.collect(sizes[-1]), want[-1])
    a.close(); b.close()
    # the depth of the pipeline: three blocks between submit and collect (three staging sets, round 6), AIDAX_ERR_STATE for a fourth;
    # collected oldest first, the same bits as the blocking path
    a, b = ax.Pool(S, 256), ax.Pool(S, 256)
    a.set_model(m); b.set_model(m)
    want = [a.process(blk) for blk in blocks[:5]]
    for k in range(3):
        b.submit(blocks[k])
    assert L.aidax_pool_submit(b.h, blocks[3].ctypes.data_as(C.POINTER(C.c_float)), sizes[3]) == -6
    for k in range(3, 5):
        assert np.array_equal(b.collect(sizes[k - 3]), want[k - 3]), k
        b.submit(blocks[k])
    for k in range(2, 5):
        assert np.array_equal(b.collect(sizes[k]), want[k]), k
    a.close(); b.close()


def test_registered_host_buffers_take_the_copies_out_of_the_pipelined_path_same_bits(tmp_path):
    """aidax_pool_register_host + aidax_pool_submit_to: the caller's own buffers, page-locked once, are the source of the
    upload and the destination of the download (no staging copies on the caller's thread) — same bits as the blocking
    path block for block, registered and unregistered buffers mixed, the calling rules enforced."""
    path = str(tmp_path / "r.json")
    modelgen.write_model(modelgen.make_model(kind="lstm", hidden=16, input_size=1, seed=23), path)
    m = ax.Model(path)
    S, n, nblk = 300, 128, 9
    x = modelgen.signal(S, n * nblk, seed=17)
    a, b = ax.Pool(S, n), ax.Pool(S, n)
    a.set_model(m); b.set_model(m)
    want = [a.process(np.ascontiguousarray(x[:, k * n:(k + 1) * n])) for k in range(nblk)]
    ins = np.empty((2, S, n), np.float32)                            # the host's two input buffers, registered as one range
    outs = [np.empty((S, n), np.float32) for _ in range(2)]          # ... its two output buffers, each a range of its own
    loose = np.empty((S, n), np.float32)                             # and one the pool has never heard of
    b.register_host(ins); b.register_host(outs[0]); b.register_host(outs[1])
    L = ax.lib()
    assert L.aidax_pool_register_host(b.h, C.c_void_p(ins.ctypes.data), ins.nbytes) == -6          # registered already
    ins[0] = x[:, :n]
    b.submit_to(ins[0], outs[0])
    for k in range(1, nblk):
        if k == 4:                                                   # an unregistered source and destination in between: staged as before
            b.submit_to(np.ascontiguousarray(x[:, k * n:(k + 1) * n]), loose)
        else:
            ins[k & 1] = x[:, k * n:(k + 1) * n]
            b.submit_to(ins[k & 1], outs[k & 1])
        prev = loose if k - 1 == 4 else outs[(k - 1) & 1]
        if k == 2:                                                   # collect must name the destination the block was submitted with
            assert L.aidax_pool_collect(b.h, loose.ctypes.data_as(C.POINTER(C.c_float)), n) == -1
        got = b.collect(n, prev)
        assert got is prev and np.array_equal(prev, want[k - 1]), k
    last = loose if nblk - 1 == 4 else outs[(nblk - 1) & 1]
    b.collect(n, last)
    assert np.array_equal(last, want[-1])
    b.unregister_host(outs[1])
    assert L.aidax_pool_unregister_host(b.h, C.c_void_p(outs[1].ctypes.data)) == -1               # not registered any more
    b.submit_to(ins[0], outs[1]); b.collect(n, outs[1])              # ... and still served, staged
    a.close(); b.close()                                             # (the pool's end releases the remaining ranges)


@pytest.mark.parametrize("kind,hidden,inputs", [("lstm", 16, 1), ("gru", 8, 2)])
def test_one_stream_pools_pass_writes_its_own_completion_word_behind_its_block(kind, hidden, inputs, tmp_path, monkeypatch):
    """The LV2 instance's pool (one stream, the block in the pool's page-locked memory): the pass of k_*_pipe / k_*_pipe4 writes the word
    aidax_pool_process polls for itself, behind a system-scope fence behind its last store — no packet follows the pass. The caller must
    never see the word before the block: 4000 blocks of fresh noise, an enabled stream bit for bit against a pool whose queue writes the word
    (AIDAX_KERNEL_WORD=0, round 5's way), then a disabled one (a raw copy, :612-619 — every sample of every block must be the one just sent),
    block lengths mixed so that consecutive blocks differ in every frame."""
    path = str(tmp_path / "w.json")
    modelgen.write_model(modelgen.make_model(kind=kind, hidden=hidden, input_size=inputs, seed=3 * hidden, in_skip=1), path)
    m = ax.Model(path)
    monkeypatch.setenv("AIDAX_KERNEL_WORD", "0")
    a = ax.Pool(1, 256)
    monkeypatch.delenv("AIDAX_KERNEL_WORD")
    b = ax.Pool(1, 256)
    a.set_model(m); b.set_model(m)
    rng = np.random.default_rng(hidden)
    sizes = [64, 256, 16, 128, 100, 32, 1, 256]
    for k in range(2000):
        n = sizes[k % len(sizes)]
        x = (rng.random((1, n), dtype=np.float32) - 0.5)
        ya, yb = a.process(x), b.process(x)
        assert np.array_equal(ya, yb), (k, n)
    b.set_controls(ax.default_controls(enabled=0.0))
    for k in range(2000):
        n = sizes[k % len(sizes)]
        x = (rng.random((1, n), dtype=np.float32) - 0.5)
        assert np.array_equal(b.process(x), x), (k, n)
    a.close(); b.close()
