"""The LV2 plugin shell (rt-neural-generic.so) driven by a mock host: descriptor and
feature handling (CPU), and on the GPU the whole plugin life cycle of the reference —
default-state restore -> worker load -> work_response swap -> NOTIFY echo -> audio,
patch:Set on CONTROL, mute-while-loading, state save, activate — compared sample for
sample with the CPU oracle's plugin mirror."""
import ctypes as C
import importlib
import os
import shutil

import numpy as np
import pytest

from oracle import oracle as O
from tests import lv2host, modelgen

THR = 1.0e-5
JSON_URI = lv2host.PLUGIN_URI.decode() + "#json"


def test_descriptor_and_symbol():
    lib = C.CDLL(lv2host.SO)
    lib.lv2_descriptor.restype = C.POINTER(lv2host.Descriptor)
    lib.lv2_descriptor.argtypes = [C.c_uint32]
    d = lib.lv2_descriptor(0)
    assert d and d.contents.URI == lv2host.PLUGIN_URI            # uris.h:26-28
    assert not lib.lv2_descriptor(1)                              # rt-neural-generic.cpp:27-28
    ext = d.contents.extension_data
    assert ext(b"http://lv2plug.in/ns/ext/worker#interface") and ext(b"http://lv2plug.in/ns/ext/state#interface")
    assert not ext(b"http://example.org/nothing")


def test_instantiate_requires_urid_map_and_worker_schedule(capfd):
    """rt-neural-generic.cpp:265-273: returns 0 when a required host feature is missing."""
    h = lv2host.Host(with_map=False)
    assert not h.handle
    h = lv2host.Host(with_schedule=False)
    assert not h.handle
    err = capfd.readouterr().err
    assert "Missing feature urid:map" in err and "Missing feature work:schedule" in err


def test_instantiate_without_gpu_fails_loudly(capfd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = lv2host.Host()
    assert not h.handle
    assert "no CPU fallback" in capfd.readouterr().err


# ----------------------------------------------------------------------------- GPU

def _oracle_controls(h):
    return O.default_controls(**{lv2host.FIELD[k]: h.ctl[k].value for k in lv2host.FIELD})


@pytest.fixture
def bundle(tmp_path):
    """A bundle directory laid out like rt-neural-generic.lv2 (models/ next to the binary)."""
    src = os.path.join(lv2host.ROOT, "tests", "golden", "models")
    dst = tmp_path / "models" / "deer ink studios"
    dst.mkdir(parents=True)
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), dst / f)
    modelgen.write_model(modelgen.make_model("gru", 16, 3, seed=7), str(tmp_path / "models" / "gru16.json"))
    return str(tmp_path)


@pytest.mark.gpu
def test_plugin_life_cycle_matches_oracle(bundle):
    h = lv2host.Host(bundle_dir=bundle)
    assert h.handle
    x = modelgen.signal(1, 256 * 14, seed=31)[0]
    blocks = [x[i * 256:(i + 1) * 256] for i in range(14)]
    plug = O.OraclePlugin()
    default_rel = "models/deer ink studios/tw40_california_clean_deerinkstudios.json"   # rt-neural-generic.ttl:315-317
    default_abs = os.path.join(bundle, default_rel)

    def step(i):
        got = h.run(blocks[i])
        want = plug.run(_oracle_controls(h), blocks[i])
        return got, want

    # 1. before any model: silence, ModelInSize 0 (loading=true, master cleared to 0)
    got, want = step(0)
    assert np.all(got == 0) and np.array_equal(got, want) and h.ctl["ModelInSize"].value == 0

    # 2. host restores the default state -> a kWorkerLoad is scheduled with the ABSOLUTE path (:741-755)
    assert h.restore(default_rel) == 0 and h.free_calls == 1
    assert len(h.work_queue) == 1
    msg = h.work_queue[0]
    assert len(msg) == 4 + 1024 and msg[4:].split(b"\0")[0].decode() == default_abs
    got, want = step(1)                                  # still silent: the worker has not run yet
    assert np.all(got == 0)

    # 3. worker thread: load; audio thread after run(): work_response swaps, echoes patch:Set, un-mutes
    assert h.pump_worker() == 1 and len(h.responses) == 1
    h.run(np.zeros(0, np.float32))                       # a pre-run with n_samples == 0 is legal (:606-609)
    plug.run(_oracle_controls(h), np.zeros(0, np.float32))
    assert h.deliver_responses() == 1
    spec = O.load_model(default_abs)
    plug.set_model(O.OracleModel(spec, 0.0, 0.0))
    note = h.read_notify()
    assert len(note) == 1 and note[0][0].endswith("patch#Set")
    props = note[0][1]
    assert props["http://lv2plug.in/ns/ext/patch#property"][1][:4] == np.uint32(h.urid(JSON_URI)).tobytes()
    vt, vb = props["http://lv2plug.in/ns/ext/patch#value"]
    assert vt.endswith("atom#Path") and vb.rstrip(b"\0").decode() == default_abs
    assert len(h.work_queue) == 1                        # kWorkerFree(old = NULL) was scheduled (:868-875)
    h.pump_worker()

    # 4. audio now follows the oracle, the master gain ramping up from 0
    for i in (2, 3, 4):
        got, want = step(i)
        assert np.abs(got - want).max() < THR, i
    assert np.abs(got).max() > 1e-3 and h.ctl["ModelInSize"].value == 1

    # 5. controls change mid-stream (EQ pre, gains)
    h.controls(EQPOS=1.0, BASS=4.0, MID=-3.0, MIDQ=1.2, TREBLE=2.0, PREGAIN=3.0, MASTER=-2.0)
    got, want = step(5)
    assert np.abs(got - want).max() < THR * 2

    # 6. patch:Set with a conditioned GRU model on CONTROL: mutes while loading (:576, :654), then swaps
    gru_abs = os.path.join(bundle, "models", "gru16.json")
    h.controls(PARAM1=0.7, PARAM2=0.2)
    h.send_patch_set(gru_abs)
    got, want6 = h.run(blocks[6]), None
    plug.set_loading(True)
    want6 = plug.run(_oracle_controls(h), blocks[6])
    assert np.abs(got - want6).max() < THR * 2
    assert len(h.work_queue) == 1 and h.pump_worker() == 1 and h.deliver_responses() == 1
    old = plug.model.ptr.contents
    plug.set_model(O.OracleModel(O.load_model(gru_abs), old.param1Coeff.target, old.param2Coeff.target))
    assert h.read_notify()[0][1]["http://lv2plug.in/ns/ext/patch#value"][1].rstrip(b"\0").decode() == gru_abs
    h.pump_worker()                                      # frees the LSTM model
    for i in (7, 8):
        got, want = step(i)
        assert np.abs(got - want).max() < THR * 2, i
    assert h.ctl["ModelInSize"].value == 3

    # 7. messages that must be ignored: wrong property, wrong value type, not a patch:Set
    h.send_patch_set(gru_abs, prop_uri="http://example.org/other")
    h.run(blocks[9]); plug.run(_oracle_controls(h), blocks[9])
    h.send_patch_set(gru_abs, value_type="http://lv2plug.in/ns/ext/atom#String")
    h.run(blocks[9]); plug.run(_oracle_controls(h), blocks[9])
    h.send_patch_set(gru_abs, otype="http://lv2plug.in/ns/ext/patch#Get")
    h.run(blocks[9]); plug.run(_oracle_controls(h), blocks[9])
    assert not h.work_queue

    # 8. a failing load leaves the plugin muted: loading stays true, the old model keeps running (:1017-1044)
    h.send_patch_set(os.path.join(bundle, "models", "missing.json"))
    got = h.run(blocks[10]); plug.set_loading(True); want = plug.run(_oracle_controls(h), blocks[10])
    assert np.abs(got - want).max() < THR * 2
    assert h.pump_worker() == 1 and not h.responses
    for i in (11, 12):
        got, want = step(i)
        assert np.abs(got - want).max() < THR * 2
    assert plug.p.masterGain.target == 0.0               # the oracle mirror is ramping to silence too (T60 0.1 s)

    # 9. state save: abstract path under #json as atom:Path, POD|PORTABLE (:783-792); no mapPath -> NO_FEATURE
    rc, stored = h.save()
    assert rc == 0 and stored == [(JSON_URI, b"models/gru16.json\0", "http://lv2plug.in/ns/ext/atom#Path", 3)]
    assert h.save(with_map_path=False)[0] == 4

    # 10. activate() clears the gain ramps to their targets (:341-342)
    h.desc.activate(h.handle); plug.activate()
    got, want = step(13)
    assert np.abs(got - want).max() < THR * 2
    h.close()


@pytest.mark.gpu
def test_disabled_port_is_a_raw_copy(bundle):
    h = lv2host.Host(bundle_dir=bundle)
    x = modelgen.signal(1, 256, seed=5)[0]
    h.controls(enabled=0.0)
    assert np.array_equal(h.run(x), x)
    h.close()


@pytest.mark.gpu
def test_hub_mode_three_instances_share_launches(bundle, monkeypatch):
    """AIDAX_HUB=4: instances that load the same model file share one pool pass per period (one period of
    latency); an instance with another file gets its own hub. Each instance against its own oracle mirror."""
    monkeypatch.setenv("AIDAX_HUB", "4")
    monkeypatch.setenv("AIDAX_HUB_DEADLINE_US", "0")      # a python host computing oracles between run() calls keeps no audio clock
    clean = os.path.join(bundle, "models/deer ink studios/tw40_california_clean_deerinkstudios.json")
    gru = os.path.join(bundle, "models", "gru16.json")
    files = [clean, clean, gru]
    hosts = [lv2host.Host(bundle_dir=bundle) for _ in files]
    assert all(h.handle for h in hosts)
    n, periods = 256, 8
    x = modelgen.signal(3, n * (periods + 2), seed=41)
    hosts[1].controls(BASS=4.0, PREGAIN=3.0)
    hosts[2].controls(PARAM1=0.6, PARAM2=0.2)
    # before any model: exact silence, like the one-stream mode
    for i, h in enumerate(hosts):
        assert np.all(h.run(x[i, :n]) == 0)
    # every instance loads its file through the worker
    for h, f in zip(hosts, files):
        h.send_patch_set(f)
        h.run(np.zeros(0, np.float32))
        h.clear_control()
        assert h.pump_worker() == 1
        assert h.deliver_responses() == 1
        h.pump_worker()
        h.run(np.zeros(0, np.float32))      # the port is written at the top of run() (:518)
        assert h.ctl["ModelInSize"].value == (3 if f == gru else 1)
    plugs = []
    for h, f in zip(hosts, files):
        p = O.OraclePlugin()
        p.set_model(O.OracleModel(O.load_model(f), 0.0, 0.0))
        p.activate()                        # a freshly attached stream starts like instantiate() + activate()
        p.set_loading(False)
        plugs.append(p)
    prev = [None] * 3
    for k in range(periods):
        blk = slice((k + 1) * n, (k + 2) * n)
        for i, h in enumerate(hosts):       # the host runs its instances one after another
            got = h.run(x[i, blk])
            if prev[i] is None:
                assert np.all(got == 0)     # first period: nothing computed yet
            else:
                assert np.abs(got - prev[i]).max() < THR * 2, (k, i, np.abs(got - prev[i]).max())
            prev[i] = plugs[i].run(_oracle_controls(h), x[i, blk])
    for h in hosts:
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [600, 777, 601])
def test_hub_mode_host_blocks_longer_than_the_hubs_go_through_in_equal_slices(bundle, monkeypatch, n):
    """Hub blocks of 256 frames, host blocks of n > 256 that 256 does not divide: the shell must cut every call into
    slices of ONE length (the hub hands a result back only to a block of the same length) — 600 -> 3 x 200, 777 -> 7 x
    111, 601 (prime) -> 601 x 1. The output is the oracle's, one slice late, with no silent stretch after the first slice.
    (Round 3 cut 601 into 256 + 256 + 89 and delivered silence for most of every block.)"""
    monkeypatch.setenv("AIDAX_HUB", "2")
    monkeypatch.setenv("AIDAX_HUB_FRAMES", "256")
    monkeypatch.setenv("AIDAX_HUB_DEADLINE_US", "0")
    clean = os.path.join(bundle, "models/deer ink studios/tw40_california_clean_deerinkstudios.json")
    h = lv2host.Host(bundle_dir=bundle, block=n)
    assert h.handle
    h.send_patch_set(clean)
    h.run(np.zeros(0, np.float32))
    h.clear_control()
    assert h.pump_worker() == 1 and h.deliver_responses() == 1
    h.pump_worker()
    h.run(np.zeros(0, np.float32))
    k = -(-n // 256)
    while n % k:
        k += 1
    L = n // k
    p = O.OraclePlugin()
    p.set_model(O.OracleModel(O.load_model(clean), 0.0, 0.0))
    p.activate()
    p.set_loading(False)
    blocks = 4
    x = modelgen.signal(1, n * blocks, seed=77)[0]
    got = np.concatenate([h.run(x[b * n:(b + 1) * n]) for b in range(blocks)])
    want = np.concatenate([p.run(_oracle_controls(h), x[o:o + L]) for o in range(0, n * blocks, L)])
    assert np.all(got[:L] == 0)                               # the first slice: nothing computed yet
    err = np.abs(got[L:] - want[:-L]).max()
    assert err < THR * 2, (n, L, err)
    assert np.abs(want[:-L]).max() > 1e-3                     # (and it is not silence that was compared)
    h.close()


@pytest.mark.gpu
def test_instances_are_placed_over_the_devices_the_process_sees(bundle, monkeypatch):
    """AIDAX_DEVICE=auto: every instance goes to the least-loaded of the hipGetDeviceCount() devices (one here: the rule itself
    is tested on CPU with injected counts, tests/test_abi_host.py); a list that names no device of this machine refuses the
    instance loudly instead of landing somewhere else. Same audio as with the default placement."""
    ax_ = importlib.import_module("aidadsp-lv2_amd")
    n_dev = ax_.device_count()
    assert n_dev >= 1
    monkeypatch.setenv("AIDAX_DEVICE", "auto")
    clean = os.path.join(bundle, "models/deer ink studios/tw40_california_clean_deerinkstudios.json")
    hosts = [lv2host.Host(bundle_dir=bundle) for _ in range(3)]
    assert all(h.handle for h in hosts)
    x = modelgen.signal(3, 512, seed=9)
    outs = []
    for i, h in enumerate(hosts):
        h.send_patch_set(clean)
        h.run(np.zeros(0, np.float32))
        h.clear_control()
        assert h.pump_worker() == 1 and h.deliver_responses() == 1
        h.pump_worker()
        outs.append(np.concatenate([h.run(x[i, :256]), h.run(x[i, 256:])]))
    p = O.OraclePlugin()
    p.set_model(O.OracleModel(O.load_model(clean), 0.0, 0.0))
    p.activate()
    p.set_loading(False)
    want = np.concatenate([p.run(_oracle_controls(hosts[0]), x[0, :256]), p.run(_oracle_controls(hosts[0]), x[0, 256:])])
    assert np.abs(outs[0] - want).max() < THR * 2
    for h in hosts:
        h.close()
    monkeypatch.setenv("AIDAX_DEVICE", str(n_dev + 3))
    h = lv2host.Host(bundle_dir=bundle)
    assert not h.handle                                  # instantiate() returns NULL, with the reason on stderr


@pytest.mark.gpu
@pytest.mark.parametrize("geo,kw", [("cfg4's stack", dict(seed=1608)), ("two cycles 1..16, two taps", dict(seed=3, conv_k=2, conv_dilations=[1, 2, 4, 8, 16] * 2))])
def test_conv_stack_through_the_plugin_at_a_hosts_block_lengths(bundle, geo, kw, monkeypatch):
    """An extension model (the reference's loader would refuse it) through the plugin's own boundary, at the block lengths a host
    really sends — run() gets the host's period (rt-neural-generic.cpp:484): blocks of 64 / 128 / 256 frames of a stack with a compiled geometry
    stream through k_conv_st, a 100-frame block and a pre-run in between go through k_conv_ms on the same state; against the oracle's plugin mirror."""
    monkeypatch.setenv("AIDAX_STRICT_REFERENCE_SET", "0")        # (the shell refuses what the reference's loader would refuse unless told otherwise: INTEGRATION §3)
    path = os.path.join(bundle, "models", "conv.json")
    modelgen.write_model(modelgen.make_model(kind="conv", hidden=16, input_size=1, in_skip=1, out_gain=-3.0, **kw), path)
    h = lv2host.Host(bundle_dir=bundle)
    plug = O.OraclePlugin()
    h.send_patch_set(path)
    h.run(np.zeros(64, np.float32)); plug.set_loading(True); plug.run(_oracle_controls(h), np.zeros(64, np.float32))
    assert h.pump_worker() == 1 and h.deliver_responses() == 1
    plug.set_model(O.OracleModel(O.load_model(path), 0.0, 0.0))
    h.pump_worker()
    sizes = [64, 64, 128, 256, 100, 0, 64, 128, 256, 256, 64]
    x = modelgen.signal(1, sum(sizes), seed=52)[0]
    pos, worst = 0, 0.0
    for i, n in enumerate(sizes):
        if i == 3:
            h.controls(EQPOS=1.0, BASS=3.0, TREBLE=-2.0, PREGAIN=2.0)
        blk = x[pos:pos + n]
        got = h.run(blk)
        want = plug.run(_oracle_controls(h), blk)
        if n:
            worst = max(worst, float(np.abs(got - want).max()))
        pos += n
    assert worst < THR, (geo, worst)
    h.close()
