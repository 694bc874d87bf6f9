"""k_mfma_lp under co-tenancy (GPU, -m gpu). The kernel runs a stream group's layers on separate workgroups that wait
for each other through a ring in global memory, which is only safe while every workgroup of its grid is resident: the
pool therefore lets ONE pool per device use it, warms models up with k_mfma (bit-identical state) and never has two of
those grids in flight; a wait that times out all the same is reported (silence for that block, AIDAX_ERR_DEVICE) and the
pool falls back to k_mfma. These tests drive exactly the situations the round-2 review named: two full-size stacked
pools at once, a full-size pool processing while another stacked model is prepared and swapped in, and the fault path."""
import ctypes as C
import importlib
import queue
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests import errlog, modelgen

pytestmark = pytest.mark.gpu

ax = importlib.import_module("aidadsp-lv2_amd")
_fp = C.POINTER(C.c_float)
OK, ERR_DEVICE = 0, -5                                                       # include/aidax.h


def _last_error():
    return ax.lib().aidax_last_error().decode(errors="replace")
CFG5 = dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2)          # BASELINE cfg5's model
CFG5_STREAMS = 2048                                                          # ... and its per-GPU stream count: 128 groups x 2 layers = 256 workgroups


def _model_file(tmp_path, name, **kw):
    j = modelgen.make_model(**kw)
    p = str(tmp_path / f"{name}.json")
    modelgen.write_model(j, p)
    return p, O.parse_model(j)


def _spread(base, S):
    idx = (np.arange(S) * 7) % base.shape[0]                                 # neighbours in a stream group carry different signals
    first = np.array([np.argmax(idx == k) for k in range(base.shape[0])])
    return base[idx], first


def test_two_full_size_stacked_pools_driven_at_once(tmp_path):
    """Two cfg5-sized pools of one process, each driven by its own thread at full speed. Only one of them may run
    k_mfma_lp (256 workgroups on 256 CUs: a second such grid would leave workgroups of both waiting for CUs that their
    spinning partners hold); the other gets k_mfma. Both match the oracle, neither reports a give-up."""
    path, spec = _model_file(tmp_path, "cfg5", **CFG5)
    m = ax.Model(path)
    block, nblk = 256, 6
    base = modelgen.signal(16, block * nblk, seed=5)
    x, first = _spread(base, CFG5_STREAMS)
    want = O.run_streams(spec, O.default_controls(), base, block)
    pools = [ax.Pool(CFG5_STREAMS, block) for _ in range(2)]
    for p in pools:
        p.set_model(m)
    assert sorted(p.kernel_name for p in pools) == ["k_chain+k_mfma", "k_mfma_ls"]          # (k_mfma_lp: the whole run() in its one launch)
    outs, failures = [None, None], []
    go = threading.Barrier(2)

    def drive(i):
        try:
            go.wait()
            got = np.empty_like(x)
            for b in range(nblk):
                got[:, b * block:(b + 1) * block] = pools[i].process(np.ascontiguousarray(x[:, b * block:(b + 1) * block]))
            pools[i].sync()                                                  # raises if a hand-over gave up
            outs[i] = got
        except Exception as e:                                               # pragma: no cover
            failures.append((i, repr(e)))

    ts = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not failures, failures
    for i in range(2):
        errlog.bound(np.abs(outs[i][first] - want).max(), 1.5e-6, "gpu_lp:two_pools")
        assert pools[i].kernel_name in ("k_chain+k_mfma", "k_mfma_ls")
    # once the holder is gone the next pool may use the kernel again
    holder = [p for p in pools if p.kernel_name == "k_mfma_ls"][0]
    holder.close()
    p3 = ax.Pool(CFG5_STREAMS, block)
    p3.set_model(m)
    assert p3.kernel_name == "k_mfma_ls"
    for p in pools + [p3]:
        p.close()


def test_full_size_pool_processes_while_stacked_models_are_prepared_and_swapped_in(tmp_path):
    """A cfg5-sized pool on k_mfma_lp keeps processing while a worker thread prepares stacked model after stacked model
    (upload + 2048-frame warm-up on the worker stream: k_mfma there, in 256-frame launches) and the audio thread swaps
    them in at block boundaries — k_mfma_lp to k_mfma_lp on a grid that fills the machine. Every block of the 16
    distinct streams against the oracle; the kernel stays k_mfma_lp; no give-up."""
    files = [_model_file(tmp_path, f"s{i}", **dict(CFG5, seed=960 + i)) for i in range(2)]
    models = [ax.Model(p) for p, _ in files]
    block, nblk = 256, 120                 # (at least 40 blocks; more while fewer than three swaps have come in on a loaded box)
    base = modelgen.signal(16, block * nblk, seed=6)
    x, first = _spread(base, CFG5_STREAMS)
    pool = ax.Pool(CFG5_STREAMS, block)
    pool.set_model(models[0])
    assert pool.kernel_name == "k_mfma_ls"
    co = O.default_controls()
    plugs = [O.OraclePlugin() for _ in range(16)]
    for p in plugs:
        p.set_model(O.OracleModel(files[0][1]))
    ready, retired = queue.Queue(maxsize=1), queue.Queue()
    stop = threading.Event()
    failure = []

    def worker():
        try:
            k = 0
            while not stop.is_set():
                while not retired.empty():
                    pool.staged_free(retired.get())
                mi = (k + 1) % 2
                sg = pool.prepare_model(models[mi])
                k += 1
                while not stop.is_set():
                    try:
                        ready.put((mi, sg), timeout=0.01)
                        break
                    except queue.Full:
                        pass
                else:
                    pool.staged_free(sg)
            while not retired.empty():
                pool.staged_free(retired.get())
        except Exception as e:                                               # pragma: no cover
            failure.append(e)

    t = threading.Thread(target=worker)
    t.start()
    swaps = 0
    try:
        for b in range(nblk):
            try:
                mi, sg = ready.get_nowait()
            except queue.Empty:
                mi = None
            if mi is not None:
                pool.commit_model(sg)
                retired.put(sg)
                swaps += 1
                for p in plugs:
                    old = p.model.ptr.contents
                    p.set_model(O.OracleModel(files[mi][1], old.param1Coeff.target, old.param2Coeff.target))
            got = pool.process(np.ascontiguousarray(x[:, b * block:(b + 1) * block]))
            assert pool.kernel_name == "k_mfma_ls"
            for k in range(16):
                want = plugs[k].run(co, base[k, b * block:(b + 1) * block])
                errlog.bound(np.abs(got[first[k]] - want).max(), 2e-6, "gpu_lp:swap_under_load")
            if b >= 39 and swaps >= 3:
                break
    finally:
        stop.set()
        t.join()
        while not ready.empty():
            pool.staged_free(ready.get()[1])
    assert not failure, failure
    assert swaps >= 3, swaps
    pool.sync()                                                              # raises if a hand-over gave up
    pool.close()


def test_a_reported_give_up_is_silence_for_that_block_and_the_pool_falls_back(tmp_path, monkeypatch):
    """The fault path end to end with the kernel's test hook (AIDAX_TUNE bit 16: workgroup 0 reports a give-up that did
    not happen, so the streams' state stays valid): the blocking entry point returns AIDAX_ERR_DEVICE and silence for
    that block, the pool serves the model with k_mfma from the next block on and matches the oracle again; the
    asynchronous entry point reports through aidax_pool_sync, once."""
    import torch
    monkeypatch.setenv("AIDAX_TUNE", "16")
    path, spec = _model_file(tmp_path, "l32x2", kind="lstm", hidden=32, input_size=1, seed=322, n_rnn=2)
    m = ax.Model(path)
    S, n, nblk = 40, 128, 5
    x = modelgen.signal(S, n * nblk, seed=8)
    want = O.run_streams(spec, O.default_controls(), x, n)
    L = ax.lib()
    pool = ax.Pool(S, n)
    pool.set_model(m)
    assert pool.kernel_name == "k_mfma_ls"
    blk = np.ascontiguousarray(x[:, :n])
    out = np.full_like(blk, 7.0)
    rc = L.aidax_pool_process(pool.h, blk.ctypes.data_as(_fp), out.ctypes.data_as(_fp), n)
    assert rc == ERR_DEVICE and "hand-over" in _last_error()
    assert not out.any()                                                     # silence, never garbage
    assert pool.kernel_name == "k_chain+k_mfma"
    for b in range(1, nblk):
        got = pool.process(np.ascontiguousarray(x[:, b * n:(b + 1) * n]))
        errlog.bound(np.abs(got - want[:, b * n:(b + 1) * n]).max(), 2e-6, "gpu_lp:after_fallback")
    pool.sync()
    pool.close()

    pool = ax.Pool(S, n)
    pool.set_model(m)
    d_in = torch.from_numpy(blk).cuda()
    d_out = torch.empty_like(d_in)
    pool.process_device(d_in.data_ptr(), d_out.data_ptr(), n)
    with pytest.raises(RuntimeError, match="hand-over"):
        pool.sync()
    pool.sync()                                                              # reported once
    assert pool.kernel_name == "k_chain+k_mfma"
    pool.process_device(d_in.data_ptr(), d_out.data_ptr(), n)
    pool.sync()
    pool.close()


def test_hub_delivers_silence_for_a_pass_that_gave_up_and_recovers(tmp_path, monkeypatch):
    """The same hook under the hub: the instance that reads the row of a faulted pass gets AIDAX_ERR_DEVICE and silence
    (never the row), later passes run on k_mfma and the audio is the oracle's again, one period late."""
    monkeypatch.setenv("AIDAX_TUNE", "16")
    path, spec = _model_file(tmp_path, "l32x2h", kind="lstm", hidden=32, input_size=1, seed=323, n_rnn=2)
    n, nblk = 128, 10
    x = modelgen.signal(1, n * nblk, seed=9)[0]
    hub = ax.Hub(4, n)
    hub.set_model(ax.Model(path))
    hub.set_deadline_us(0)
    slot = hub.attach()
    plug = O.OraclePlugin()
    plug.set_model(O.OracleModel(spec))
    co = O.default_controls()
    L = ax.lib()
    wants, errors, good = [], 0, 0
    for b in range(nblk):
        blk = np.ascontiguousarray(x[b * n:(b + 1) * n])
        wants.append(plug.run(co, blk))
        out = np.full(n, 7.0, np.float32)
        rc = L.aidax_hub_run(hub.h, slot, blk.ctypes.data_as(_fp), out.ctypes.data_as(_fp), n)
        if rc != OK:
            assert rc == ERR_DEVICE and not out.any()
            errors += 1
            assert good == 0                                                 # faults only at the start: the fallback holds
        elif b >= 1 and out.any():
            errlog.bound(np.abs(out - wants[b - 1]).max(), 2e-6, "gpu_lp:hub_after_fallback")
            good += 1
    assert 1 <= errors <= 3 and good >= nblk - 5, (errors, good)
    hub.close()


def test_a_chained_grid_that_cannot_be_resident_is_refused_at_launch_and_served_by_k_mfma(tmp_path, monkeypatch):
    """Residency as a launch-time guarantee: the chained kernels go out through hipLaunchCooperativeKernel, which refuses a
    grid that cannot be co-resident on the device — here a pool FORCED onto the kernel (AIDAX_MFMA_LP=1) with more
    (stream group, layer) workgroups than the GPU has CUs. No launch, no wait, no give-up: the same call serves the block
    with k_mfma, every block matches the oracle, and the pool stays on k_mfma. (With AIDAX_LP_COOP=0 the same pool launches
    the grid and leans on the dispatcher's order — the situation the 250 ms give-up exists for.)"""
    import torch
    monkeypatch.setenv("AIDAX_MFMA_LP", "1")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # (LSTM-96 x 2: a workgroup's 152 KiB of LDS make it one per CU — a narrow model's grid of this size WOULD fit, several per CU)
    path, spec = _model_file(tmp_path, "l96x2big", kind="lstm", hidden=96, input_size=1, seed=965, n_rnn=2)
    S, n, nblk = 16 * (cus // 2 + 3), 128, 3                                   # (cus / 2 + 3) groups x 2 layers > cus workgroups
    base = modelgen.signal(8, n * nblk, seed=12)
    idx = (np.arange(S) * 5) % 8
    pool = ax.Pool(S, n)
    pool.set_model(ax.Model(path))
    assert pool.kernel_name == "k_mfma_ls"
    want = O.run_streams(spec, O.default_controls(), base, n)
    for b in range(nblk):
        got = pool.process(np.ascontiguousarray(base[idx, b * n:(b + 1) * n]))
        assert pool.kernel_name == "k_chain+k_mfma"
        for k in range(8):
            errlog.bound(np.abs(got[int(np.argmax(idx == k))] - want[k, b * n:(b + 1) * n]).max(), 2e-6, "gpu_lp:refused_grid")
    pool.sync()                                                                # nothing gave up: nothing was launched
    pool.close()
