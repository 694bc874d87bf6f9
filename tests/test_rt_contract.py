"""The threading contract of the drop-in boundary (rt-neural-generic.cpp:807-893 and the
`lv2:hardRTCapable` claim of rt-neural-generic.ttl:25): everything heavy happens in work() on the worker
thread; work_response() and run() on the audio thread neither allocate nor free device or pinned memory, nor
wait for the device. Proven from outside the library by an LD_PRELOAD interposer on the HIP runtime
(tests/hip_audit.c) around a full plugin life cycle (tests/rt_audit.py), plus the audio result of a swap that is
prepared on one thread while another keeps playing."""
import importlib
import json
import os
import shutil
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests import modelgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ax = importlib.import_module("aidadsp-lv2_amd")
THR = 1.0e-5

FORBIDDEN_ON_AUDIO_THREAD = ("hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipDeviceSynchronize",
                             "hipMemcpy", "hipStreamCreate", "hipEventCreate")


@pytest.fixture
def bundle(tmp_path):
    src = os.path.join(ROOT, "tests", "golden", "models")
    dst = tmp_path / "models" / "deer ink studios"
    dst.mkdir(parents=True)
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), dst / f)
    modelgen.write_model(modelgen.make_model("gru", 16, 3, seed=7), str(tmp_path / "models" / "gru16.json"))
    modelgen.write_model(modelgen.make_model("lstm", 32, 1, seed=32), str(tmp_path / "models" / "lstm32.json"))
    return str(tmp_path)


def test_audit_shim_builds(tmp_path):
    so = str(tmp_path / "libhipaudit.so")
    cc = subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-Wall", "-Wextra", "-Werror",
                         os.path.join(ROOT, "tests", "hip_audit.c"), "-o", so, "-ldl"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr


@pytest.mark.gpu
def test_audio_thread_entry_points_do_not_allocate_or_wait_for_the_device(tmp_path, bundle):
    so = str(tmp_path / "libhipaudit.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", os.path.join(ROOT, "tests", "hip_audit.c"), "-o", so, "-ldl"], check=True)
    env = dict(os.environ, LD_PRELOAD=so, HIP_AUDIT_LIB=so, AIDAX_NO_TORCH="1")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rt_audit.py"), bundle], env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    rep = json.loads(run.stdout.strip().splitlines()[-1])
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rt_audit.json"), "w") as f:
            json.dump(rep, f, indent=1)

    # the interposer sees the library's calls: the worker half allocates, uploads and waits
    for phase in ("work_load", "abi_prepare"):
        assert rep[phase]["hipMalloc"] >= 3 * rep[phase]["calls"] and rep[phase]["hipStreamSynchronize"] >= 1, rep[phase]
    assert rep["work_free"]["hipFree"] >= 1 and rep["abi_staged_free"]["hipFree"] >= 2

    # audio thread: nothing forbidden, anywhere
    audio = ("run_no_model", "work_response", "run_model", "run_controls_changed", "run_patch_set", "activate",
             "run_pre_run", "abi_commit", "abi_set_controls", "abi_process", "hub_run")
    for phase in audio:
        for call in FORBIDDEN_ON_AUDIO_THREAD:
            assert rep[phase][call] == 0, (phase, call, rep[phase])
    # work_response() / commit / activate / set_controls: no wait of any kind, a handful of async operations
    for phase in ("work_response", "abi_commit", "activate", "abi_set_controls"):
        assert rep[phase]["hipStreamSynchronize"] == 0 and rep[phase]["hipEventSynchronize"] == 0, (phase, rep[phase])
        assert rep[phase]["launch"] <= 2 * rep[phase]["calls"], (phase, rep[phase])
    # run(): exactly one completion per call, for the stream that carries its block — a word in pinned host memory that the caller
    # polls (small pools), or one hipStreamSynchronize. With a model playing, the pass of a one-stream pool writes that word ITSELF behind
    # its block (k_*_pipe / k_*_pipe4: nothing follows the pass on its queue, round 6); without one, and for the pre-run call of no
    # frames, the stream writes it behind the pass.
    for phase in ("run_model", "run_controls_changed", "run_patch_set"):
        assert rep[phase]["hipStreamSynchronize"] == 0 and rep[phase]["hipStreamWriteValue32"] == 0, (phase, rep[phase])
        assert rep[phase]["launch"] == rep[phase]["calls"], (phase, rep[phase])
    for phase in ("run_no_model", "run_pre_run", "abi_process"):
        assert rep[phase]["hipStreamSynchronize"] + rep[phase]["hipStreamWriteValue32"] == rep[phase]["calls"], (phase, rep[phase])
    for phase in ("run_no_model", "run_model", "run_controls_changed", "run_patch_set", "abi_process", "run_pre_run"):
        assert rep[phase]["hipEventSynchronize"] == 0, (phase, rep[phase])
    # hub mode: run() of an instance launches nothing and copies nothing through the runtime (the launcher thread does)
    assert rep["hub_run"]["launch"] == 0 and rep["hub_run"]["hipMemcpyAsync"] == 0 and rep["hub_run"]["hipStreamSynchronize"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 96, 1024])
def test_swap_prepared_on_a_worker_thread_while_the_old_model_plays(tmp_path, streams):
    """prepare on a second thread while the audio side keeps processing blocks of the old model; commit between
    two blocks. Expected audio: the oracle plugin whose model is replaced at that block boundary, the new model's
    PARAM smoothers built around the targets in force when the worker ran (rt-neural-generic.cpp:822-825)."""
    ja = modelgen.make_model("lstm", 16, 2, seed=11)
    jb = modelgen.make_model("gru", 24, 3, seed=12)
    pa, pb = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    modelgen.write_model(ja, pa); modelgen.write_model(jb, pb)
    ma, mb = ax.Model(pa), ax.Model(pb)
    n, blocks = 128, 10
    pool = ax.Pool(streams, n)
    pool.set_model(ma)
    ctl = dict(param1=0.6, param2=0.25, bass_boost_db=2.0)
    pool.set_controls(ax.default_controls(**ctl))
    x = modelgen.signal(streams, n * blocks, seed=77)
    check = sorted(set([0, streams // 2, streams - 1]))
    c = O.default_controls(**ctl)
    staged = {}
    started, go = threading.Event(), threading.Event()

    def worker():
        started.set()
        go.wait()
        staged["sg"] = pool.prepare_model(mb)

    t = threading.Thread(target=worker); t.start(); started.wait()
    got = []
    for b in range(blocks):
        if b == 3:
            go.set()                               # the worker prepares while blocks 3.. keep playing model A
        if b == 6:
            t.join()
            pool.commit_model(staged["sg"])        # audio thread, between two blocks
        got.append(pool.process(np.ascontiguousarray(x[:, b * n:(b + 1) * n])))
    pool.staged_free(staged["sg"])
    got = np.concatenate(got, axis=1)
    for s in check:
        p = O.OraclePlugin(); p.set_model(O.OracleModel(O.parse_model(ja)))
        want = []
        for b in range(blocks):
            if b == 6:
                old = p.model.ptr.contents
                p.set_model(O.OracleModel(O.parse_model(jb), old.param1Coeff.target, old.param2Coeff.target))
            want.append(p.run(c, x[s, b * n:(b + 1) * n]))
        want = np.concatenate(want)
        assert np.abs(got[s] - want).max() < THR, (s, np.abs(got[s] - want).max())
    pool.close()


@pytest.mark.gpu
def test_failed_prepare_leaves_the_pool_untouched(tmp_path):
    """a model the pool cannot host (stacked model on an 8192-frame pool) fails in prepare — on the worker —
    and the playing model is unaffected; an unload is a prepared empty slot."""
    ja = modelgen.make_model("lstm", 12, 1, seed=5)
    jx = modelgen.make_model("lstm", 96, 1, seed=6, n_rnn=2)
    pa, px = str(tmp_path / "a.json"), str(tmp_path / "x.json")
    modelgen.write_model(ja, pa); modelgen.write_model(jx, px)
    pool = ax.Pool(2, 8192)
    pool.set_model(ax.Model(pa))
    x = modelgen.signal(2, 512, seed=9)
    plug = O.OraclePlugin(); plug.set_model(O.OracleModel(O.parse_model(ja)))
    c = O.default_controls()
    y0 = pool.process(np.ascontiguousarray(x[:, :256]))
    with pytest.raises(ax.AidaxError):
        pool.prepare_model(ax.Model(px))
    y1 = pool.process(np.ascontiguousarray(x[:, 256:]))
    want = np.concatenate([plug.run(c, x[0, :256]), plug.run(c, x[0, 256:])])
    assert np.abs(np.concatenate([y0[0], y1[0]]) - want).max() < THR
    sg = pool.prepare_model(None)                  # unload
    pool.commit_model(sg)
    pool.staged_free(sg)
    assert pool.kernel_name == "k_nomodel"
    pool.close()
