"""CPU-only tests of the C-ABI library: it loads, exports every symbol the header
declares, and its host-side logic (json loader, architecture predicate, biquad
design, control reduction) matches the reference behaviour. No compute call is
made here; device calls must fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import importlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import modelgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ax = importlib.import_module("aidadsp-lv2_amd")


def test_library_loads_and_exports_every_declared_symbol():
    L = ax.lib()
    names = ax.declared_symbols()
    assert len(names) >= 25, names
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/aidax.h but not exported"
    assert b"gfx950" in L.aidax_version()


def test_no_torch_types_and_no_oracle_in_the_product():
    """The boundary is plain C; the product never links or opens the oracle."""
    with open(os.path.join(ROOT, "include", "aidax.h")) as f:
        hdr = f.read()
    assert "torch" not in hdr and "at::" not in hdr
    from tests.conftest import HOOKS_LIB, SHIP_LIB
    for path in (SHIP_LIB, HOOKS_LIB):
        out = os.popen(f"ldd '{path}'").read()
        assert "oracle" not in out and "libamdhip64" in out
    csrc = os.path.join(ROOT, "aidadsp-lv2_amd", "csrc")
    for fn in os.listdir(csrc):
        with open(os.path.join(csrc, fn)) as f:
            assert "oracle/" not in f.read().replace("vs the CPU oracle", ""), fn


def test_bundled_models_load_like_the_reference(bundled_models):
    for path in bundled_models:
        m = ax.Model(path)
        i = m.info
        assert (i.cell, i.hidden, i.input_size, i.n_rnn_layers, i.input_skip) == (0, 12, 1, 1, 0)
        assert i.input_gain == 1.0 and i.output_gain == 1.0
        assert i.samplerate == 48000.0          # metadata.samplerate is the STRING "48000" -> default (:1005-1013)
        assert i.n_golden == 2048 and i.in_reference_set == 1
        assert i.n_weights == 685               # LSTM-12 685 floats
        assert m.path == path
        gi, go = m.golden()
        spec = O.load_model(path)
        assert np.array_equal(gi, spec.input_batch) and np.array_equal(go, spec.output_batch)


@pytest.mark.parametrize("cell", ["lstm", "gru"])
@pytest.mark.parametrize("hidden", modelgen.HIDDEN_SIZES)
def test_all_54_reference_variants_are_accepted(cell, hidden):
    """variant/generate_variant_hpp.py:3-6 — {GRU,LSTM} x 9 hidden x 3 input sizes."""
    for isz in modelgen.INPUT_SIZES:
        j = modelgen.make_model(cell, hidden, isz, seed=hidden * 10 + isz)
        m = ax.Model(text=json.dumps(j))
        i = m.info
        assert (i.cell, i.hidden, i.input_size, i.in_reference_set) == (0 if cell == "lstm" else 1, hidden, isz, 1)


def test_loader_key_semantics():
    base = modelgen.make_model("lstm", 16, 2, seed=1)
    m = ax.Model(text=json.dumps(dict(base, in_skip=1, in_gain=-6.0, out_gain=3.0, samplerate=44100)))
    i = m.info
    assert i.input_skip == 1 and i.samplerate == 44100.0
    assert i.input_gain == pytest.approx(O.db_co(-6.0), abs=0) and i.output_gain == pytest.approx(O.db_co(3.0), abs=0)
    # non-numeric values are ignored, not errors (:982-1003)
    i = ax.Model(text=json.dumps(dict(base, in_skip="1", in_gain="x", out_gain=None))).info
    assert (i.input_skip, i.input_gain, i.output_gain) == (0, 1.0, 1.0)
    # metadata.samplerate as a NUMBER wins over top-level samplerate (:1005-1010)
    j = dict(base, samplerate=44100)
    j["metadata"] = {"samplerate": 96000}
    assert ax.Model(text=json.dumps(j)).info.samplerate == 96000.0


def test_loader_errors_are_codes_not_exceptions(tmp_path):
    base = modelgen.make_model("lstm", 16, 1, seed=2)
    L = ax.lib()

    def code(text):
        h = C.c_void_p()
        rc = L.aidax_model_load_memory(text.encode(), len(text), b"t", C.byref(h))
        assert not h.value or rc == 0
        return rc, L.aidax_last_error().decode()

    assert code("{ not json")[0] == -3
    rc, msg = code(json.dumps({k: v for k, v in base.items() if k != "in_shape"}))
    assert rc == -3 and "Unable to load json file" in msg
    assert code(json.dumps(dict(base, in_shape=[None, None, 4])))[0] == -4          # input_size > MAX_INPUT_SIZE (:978-980)
    assert code(json.dumps(dict(base, in_skip=2)))[0] == -4                           # in_skip > 1 (:984-985)
    rc, msg = code(json.dumps(modelgen.make_model("lstm", 28, 1)))                    # 28 not in the size list
    assert rc == -4 and "Unable to identify a known model architecture" in msg
    bad = json.loads(json.dumps(base))
    bad["layers"][0]["type"] = "rnn"
    assert code(json.dumps(bad))[0] == -4
    bad = json.loads(json.dumps(base))
    bad["layers"][0]["weights"][1] = bad["layers"][0]["weights"][1][:-1]
    assert code(json.dumps(bad))[0] == -3
    h = C.c_void_p()
    assert L.aidax_model_load(str(tmp_path / "nope.json").encode(), C.byref(h)) == -2
    assert L.aidax_model_load(None, C.byref(h)) == -1


def test_biquad_design_bit_exact_vs_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "dsp_ref.npz"))
    for (t, fc, q, gain), want in zip(g["designs"], g["coeffs"]):
        got = ax.biquad_design(int(t), fc, q, gain)
        assert np.array_equal(got, want), (t, fc, q, gain)


def test_biquad_design_bit_exact_vs_oracle_random():
    rs = np.random.RandomState(5)
    L = O.lib()
    for _ in range(2000):
        t = int(rs.randint(0, 7))
        fc, q, gain = float(rs.uniform(1e-4, 0.499)), float(rs.uniform(0.2, 5)), float(rs.uniform(-15, 15))
        f = O.Biquad()
        L.orc_biquad_init(C.byref(f), t, fc, q, gain)
        assert np.array_equal(ax.biquad_design(t, fc, q, gain), [f.a0, f.a1, f.a2, f.b1, f.b2])


def test_db_and_lpf_maps_match_oracle():
    for db in np.linspace(-100, 20, 241, dtype=np.float32):
        assert ax.db_to_coeff(float(db)) == O.db_co(float(db))
    assert ax.db_to_coeff(-90.0) == 0.0 and ax.db_to_coeff(-89.99) > 0.0
    L = O.lib()
    for pc in np.linspace(0, 100, 101, dtype=np.float32):
        assert ax.lpf_fc(float(pc)) == float(L.orc_lpf_fc(C.c_float(float(pc))))


def test_control_defaults_match_ttl():
    c, o = ax.default_controls(), O.default_controls()
    for n in O.CONTROL_FIELDS:
        assert getattr(c, n) == getattr(o, n), n



def test_conv_stack_form_rule_by_shape(tmp_path):
    """aidax_model_conv_form — which kernel family a conv1d stack runs on is a pure function of its shape (the packer's rules): the
    sixteen-channel stacks whose geometry is compiled (aidax_layout.h StGeoA ..: BASELINE cfg4's eight layers of three taps with dilation
    2^l; ten layers of two taps in two cycles 1 .. 16; six layers of three taps; two cycles 1 .. 8 of three taps; cfg4's dilations with two
    taps) stream their blocks of 64 / 128 / 256 frames through k_conv_st (4); other sixteen-channel stacks of two to four taps stay
    layer-major on k_conv_ms (3); narrower or wider-tap stacks on the fp32 matrix kernel (2); recurrent models are not conv stacks (0)."""
    from tests import modelgen
    ax = importlib.import_module("aidadsp-lv2_amd")

    def form(**kw):
        return ax.Model(modelgen.write_model(modelgen.make_model(**kw), str(tmp_path / "m.json"))).conv_form
    assert form(kind="conv", hidden=16, input_size=1, seed=1608) == 4                       # cfg4
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_k=2, conv_dilations=[1, 2, 4, 8, 16] * 2) == 4
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=6) == 4
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_dilations=[1, 2, 4, 8] * 2) == 4
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=8, conv_k=2) == 4
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=5) == 3           # cfg4's taps and dilations, five layers: no geometry
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_layers=7, conv_k=4) == 3 # seven layers of four taps
    assert form(kind="conv", hidden=16, input_size=1, seed=3, conv_k=2, conv_dilations=[1, 2, 4, 8, 16, 1, 2, 4, 8, 32]) == 3
    assert form(kind="conv", hidden=8, input_size=1, seed=85, conv_layers=4, conv_k=5) == 2 # eight channels, five taps
    assert form(kind="lstm", hidden=32, input_size=1, seed=32) == 0
    # dilations that are not 2^l: an eight-layer stack of three taps that k_conv_st does not take
    rs = np.random.RandomState(5)
    layers = [modelgen.conv_layer(rs, 1, 16, 3, 1, "tanh")] + [modelgen.conv_layer(rs, 16, 16, 3, 3 * (l + 1), "tanh") for l in range(7)] + [modelgen.dense_layer(rs, 16)]
    j = {"in_shape": [None, None, 1], "layers": layers, "metadata": {"name": "odd", "samplerate": "48000"}, "in_skip": 0, "in_gain": 0.0, "out_gain": 0.0}
    assert ax.Model(modelgen.write_model(j, str(tmp_path / "odd.json"))).conv_form == 3


def test_launch_form_rule_for_one_layer_table_models(monkeypatch):
    """aidax_many_streams_form — the decision table that sends a pool of a one-layer table model to k_quad (1) or to the
    16-stream matrix-core kernels (2) — is a pure function of (cell, hidden, streams, CUs): the measured crossovers of
    profiles/r03_cfg3_forms.txt, r04_gs_threshold.txt and r04_ls1_ab.txt pinned on CPU, the thresholds moving with the CU count
    (they are counted in rounds of workgroups), the environment switches taking effect per call."""
    ax = importlib.import_module("aidadsp-lv2_amd")
    for k in ("AIDAX_LSTM_GS", "AIDAX_LS1", "AIDAX_LP_SPLIT", "AIDAX_GRU_GM"):
        monkeypatch.delenv(k, raising=False)
    f, LSTM, GRU = ax.many_streams_form, 0, 1
    # small cells: per-stream forms up to 4095 streams, k_quad from 4096; 32 units on the split kernel beyond 6144
    assert [f(LSTM, 16, n) for n in (1, 1024, 4095, 4096, 16384)] == [0, 0, 0, 1, 1]
    assert [f(GRU, 32, n) for n in (2048, 4096, 6144, 6145, 16384)] == [0, 2, 1, 2, 2]          # (4096: one full round of k_mfma_lp's one-launch form)
    # LSTM-64: k_quad up to 1024 streams, k_lstm_gs beyond; LSTM-40: k_lstm_gs from 2049 to 4096, k_mfma_ls1 beyond
    assert [f(LSTM, 64, n) for n in (64, 1024, 1025, 4096, 16384)] == [1, 1, 2, 2, 2]
    assert [f(LSTM, 40, n) for n in (256, 512, 2048, 2049, 4096, 8192)] == [0, 1, 1, 2, 2, 2]
    # GRU-64 on k_gru_gs from sixteen stream groups (241 streams), GRU-40 from 2049; the 80-unit cells on the matrix cores beyond 1024 streams
    assert [f(GRU, 64, n) for n in (1, 240, 241, 4096)] == [2, 2, 2, 2]    # round 5: at every size, the one-stream pool included (profiles/r05_blocklen_forms3.txt)
    assert [f(GRU, 40, n) for n in (2048, 2049)] == [0, 2]
    assert [f(GRU, 80, n) for n in (1024, 1025)] == [1, 2] and [f(LSTM, 80, n) for n in (1024, 1025)] == [1, 2]
    # round 5, the table at the block lengths a host really uses (rt-neural-generic.cpp:484: run() gets the host's period): LSTM-32 between
    # one and one and a half rounds of stream groups belongs to the split form (k_nn), not to k_quad; from 352 groups (5617 streams) and
    # 192-frame blocks to k_mfma_ls1 — the one crossover that moves with the block length
    assert [f(LSTM, 32, n) for n in (4096, 4097, 5120, 5616, 5617, 6144, 6145)] == [2, 0, 0, 0, 2, 2, 2]
    assert [f(LSTM, 32, n, 256, 64) for n in (4096, 4097, 5120, 5616, 5617, 6144, 6145)] == [2, 0, 0, 0, 0, 0, 2]
    assert [f(LSTM, 32, n, 256, 128) for n in (5617, 6144, 6145)] == [0, 0, 2] and [f(LSTM, 32, n, 256, 192) for n in (5617, 6144)] == [2, 2]
    assert [f(GRU, 64, n, 256, 64) for n in (1, 64, 4096)] == [2, 2, 2] and [f(LSTM, 64, n, 256, 64) for n in (1024, 1025)] == [1, 2]
    # half the CUs (a partitioned device): the stream counts of the crossovers halve
    assert [f(LSTM, 64, n, 128) for n in (512, 513)] == [1, 2] and [f(GRU, 80, n, 128) for n in (512, 513)] == [1, 2]
    # round 6: EVERY threshold is counted in rounds of sixteen-stream groups — a 64-CU and a 32-CU partition (CPX mode) see the whole table
    # at a quarter / an eighth of the stream counts it was measured at (256 CUs: 4096 streams = one round)
    for cus in (64, 32):
        k = 256 // cus
        assert [f(LSTM, 16, n, cus) for n in (1, 4095 // k, 4096 // k, 16384 // k)] == [0, 0, 1, 1], cus
        assert [f(GRU, 32, n, cus) for n in (2048 // k, 4096 // k, 6144 // k, 6144 // k + 1, 16384 // k)] == [0, 2, 1, 2, 2], cus
        assert [f(LSTM, 64, n, cus) for n in (64 // k, 1024 // k, 1024 // k + 1, 4096 // k)] == [1, 1, 2, 2], cus
        assert [f(LSTM, 40, n, cus) for n in (256 // k, 512 // k, 2048 // k, 2048 // k + 1, 8192 // k)] == [0, 1, 1, 2, 2], cus
        assert [f(GRU, 40, n, cus) for n in (2048 // k, 2048 // k + 1)] == [0, 2], cus
        assert [f(LSTM, 32, n, cus) for n in (4096 // k, 4096 // k + 1, 5120 // k, 6144 // k, 6144 // k + 1)] == [2, 0, 0, 2, 2], cus
    # a pool for blocks shorter than any measured (hub mode's four-frame placeholder pool): GRU-64 keeps round 4's threshold there
    assert [f(GRU, 64, n, 256, 4) for n in (1, 240, 241, 4096)] == [1, 1, 2, 2] and [f(GRU, 64, n, 256, 64) for n in (1, 240)] == [2, 2]
    # switches: read per call
    monkeypatch.setenv("AIDAX_LSTM_GS", "0")
    assert f(LSTM, 64, 2048) == 1 and f(LSTM, 64, 2049) == 2          # k_mfma_ls1's rule takes over (beyond 2048 streams)
    monkeypatch.setenv("AIDAX_LP_SPLIT", "0")
    assert f(LSTM, 64, 3584) == 1 and f(LSTM, 64, 3585) == 2          # round 3's table: k_mfma_lp where its one round is at least seven eighths full


def test_placement_rule_picks_the_least_loaded_candidate():
    """aidax_pick_device — how the LV2 shell spreads instances / hubs over the GPUs of a node (one instance = one stream,
    rt-neural-generic.cpp:244-333) — is a pure function: the device count and the loads are injected here."""
    ax = importlib.import_module("aidadsp-lv2_amd")
    pick = ax.pick_device
    assert pick(None, 8) == 0 and pick("", 8) == 0 and pick("0", 8) == 0            # AIDAX_DEVICE unset: device 0, as before
    assert pick("auto", 8) == 0                                                      # all idle: lowest index
    assert pick("auto", 8, [3, 1, 2, 1, 5, 9, 9, 9]) == 1                            # least loaded, ties to the lowest index
    assert pick("auto", 8, [1, 1, 1, 1, 1, 1, 1, 0]) == 7
    assert pick("2-5", 8, [0, 0, 4, 3, 3, 9, 0, 0]) == 3
    assert pick("6,1,3", 8, [0, 2, 0, 2, 0, 0, 1, 0]) == 6
    assert pick("0-3,6", 8, [5, 5, 5, 5, 0, 0, 4, 0]) == 6
    assert pick("4-15", 8, [0, 0, 0, 0, 2, 1, 2, 2]) == 5                            # the part of a range the machine has
    assert pick("3", 1 + 3) == 3
    # round-robin emerges from least-loaded: sixteen instances over eight devices, two each
    load = [0] * 8
    for _ in range(16):
        load[pick("auto", 8, load)] += 1
    assert load == [2] * 8
    for bad, n in (("8", 8), ("9-12", 8), ("x", 8), ("1,,2", 8), ("3-1", 8), ("1-", 8), ("auto", 0), ("0,", 8), ("-1", 8)):
        with pytest.raises(ax.AidaxError):
            pick(bad, n)

def test_hub_rule_keeps_a_playing_instance_on_its_device():
    """aidax_pick_hub with injected devices (no GPU involved): a model-file swap of an instance that already plays must land on the
    device it plays on — work_response() carries the biquad memories and gain smoothers seat to seat with a device-side copy
    (rt-neural-generic.cpp:868-875 keeps them), which pools on two devices cannot do — whatever the loads say; a FIRST join spreads."""
    ax = importlib.import_module("aidadsp-lv2_amd")
    pick = ax.pick_hub
    # first join, no hub of the file yet: a new hub on the least-loaded candidate
    assert pick([], [], -1, "auto", 4, [2, 1, 0, 3]) == (-1, 2)
    assert pick([], [], -1, None, 4, [9, 0, 0, 0]) == (-1, 0)                       # AIDAX_DEVICE unset: device 0
    # first join, hubs of the file exist: the first with a free seat, wherever it is
    assert pick([1, 3], [0, 2], -1, "auto", 4, [0, 5, 0, 5]) == (1, 3)
    assert pick([1, 3], [0, 0], -1, "auto", 4, [4, 5, 1, 5]) == (-1, 2)             # all full: a new hub, least-loaded device
    # a playing instance (device 1): hubs elsewhere do not count, however idle those devices are
    assert pick([0, 2, 1], [8, 8, 1], 1, "auto", 4, [0, 9, 0, 0]) == (2, 1)
    assert pick([0, 2, 1], [8, 8, 0], 1, "auto", 4, [0, 9, 0, 0]) == (-1, 1)        # its device's hub is full: a NEW hub there
    assert pick([0, 2], [8, 8], 1, "auto", 4, [0, 9, 0, 0]) == (-1, 1)              # the file has no hub on its device yet
    assert pick([], [], 3, "0", 4) == (-1, 3)                                       # ... even when AIDAX_DEVICE now names another
    # twenty swaps of instances spread over four devices: nobody ever changes device
    hubs_dev, hubs_free = [0, 1, 2, 3], [1, 1, 1, 1]
    for k in range(20):
        cur = k % 4
        idx, dev = pick(hubs_dev, hubs_free, cur, "auto", 4, [k, 0, 0, 0])
        assert dev == cur and (idx == -1 or hubs_dev[idx] == cur)
        if idx >= 0:
            hubs_free[idx] -= 1
        else:
            hubs_dev.append(dev)
            hubs_free.append(3)
    with pytest.raises(ax.AidaxError):
        pick([], [], 4, "auto", 4)                                                  # current device beyond the machine
    with pytest.raises(ax.AidaxError):
        pick([], [], -1, "7", 4)                                                    # a first join still needs a valid AIDAX_DEVICE


def test_device_calls_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ax.AidaxError) as e:
        ax.Pool(4, 256)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_header_is_plain_c_and_a_c_program_links(tmp_path, bundled_models):
    """include/aidax.h compiles as C99 with -pedantic; a C program loads a model, and without a GPU
    aidax_pool_create reports AIDAX_ERR_DEVICE (-5) instead of falling back to anything."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    libdir = os.path.dirname(ax.lib_path())
    exe = str(tmp_path / "c_abi_smoke")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c_abi_smoke.c")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    cc = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, src, "-o", exe,
                         "-L", libdir, "-laidax_hip", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    model = [m for m in bundled_models if "california_clean" in m][0]
    run = subprocess.run([exe, model], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "cell=0 hidden=12 inputs=1 layers=1" in run.stdout
    assert "null handles: submit=-1 collect=-1 export=-1 import=-1 successor=-1 adopt=-1 hub_frames=0" in run.stdout
    assert "sizeof(aidax_stream_dsp)=136" in run.stdout          # 7 x 2 doubles + 6 floats: the binding's ctypes mirror
    import torch
    if not torch.cuda.is_available():
        assert "pool_create rc=-5 pool=null" in run.stdout


def test_the_shipped_library_has_no_test_or_measurement_switch():
    """Two builds of one source tree (Makefile, aidax_layout.h: AIDAX_TEST_HOOKS): the shipped library — bench.py, smoke(), the LV2
    shell, the bundle — reads only its documented configuration from the environment; every switch that forces a kernel form, injects a
    fault or selects a measurement path exists only in lib/hooks/, which this suite loads (tests/conftest.py)."""
    import re
    import subprocess
    from tests.conftest import HOOKS_LIB, SHIP_LIB

    def names(path):
        out = subprocess.run(["strings", "-a", path], capture_output=True, text=True, check=True).stdout
        return set(re.findall(r"AIDAX_[A-Z0-9_]+", out))
    ship, hooks = names(SHIP_LIB), names(HOOKS_LIB)
    assert ship == {"AIDAX_SPIN_WAIT", "AIDAX_ZEROCOPY", "AIDAX_STRICT_REFERENCE_SET", "AIDAX_KEEP_WARM_US", "AIDAX_KERNEL_WORD"}, ship      # configuration, INTEGRATION.md §3
    assert {"AIDAX_TUNE", "AIDAX_KERNEL", "AIDAX_LP_COOP", "AIDAX_MFMA_LP", "AIDAX_LP_SPLIT"} <= hooks
    shell = names(os.path.join(ROOT, "aidadsp-lv2_amd", "lv2", "rt-neural-generic.so"))
    assert shell == {"AIDAX_DEVICE", "AIDAX_HUB", "AIDAX_HUB_FRAMES", "AIDAX_HUB_DEADLINE_US", "AIDAX_STRICT_REFERENCE_SET"}, shell
    from tests.conftest import SHIP_LEG
    assert os.path.samefile(ax.lib_path(), SHIP_LIB if SHIP_LEG else HOOKS_LIB)          # what this process runs on
    # both export everything include/aidax.h declares
    import ctypes
    for path in (SHIP_LIB, HOOKS_LIB):
        L = ctypes.CDLL(path)
        assert not [n for n in ax.declared_symbols() if not hasattr(L, n)], path
