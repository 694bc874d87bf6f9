"""The host-only sources of the library (json parser + loader, weight packers, control-rate DSP design) under
AddressSanitizer + UndefinedBehaviorSanitizer (CPU suite only; the GPU box has no sanitizer runs). `make asan` builds
tests/asan_harness.cpp with the product's own aidax_model.cpp / json_min.h / aidax_pack.cpp / aidax_dsp_host.cpp;
the corpus is every architecture of the reference's table (54 variants), the extension models, and the mutated files
of tests/test_loader_fuzz.py written to disk."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import modelgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    r = subprocess.run(["make", "-s", "-C", ROOT, "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return os.path.join(ROOT, "build", "asan", "asan_harness")


def _run(harness, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([harness] + files, capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    return r.stdout


def test_every_architecture_through_loader_and_packers_under_sanitizers(harness, tmp_path):
    files = []
    for kind in ("lstm", "gru"):
        for hidden in modelgen.HIDDEN_SIZES:
            for isz in (1, 2, 3):
                p = str(tmp_path / f"{kind}{hidden}_{isz}.json")
                modelgen.write_model(modelgen.make_model(kind, hidden, isz, seed=hidden + isz, in_skip=isz & 1), p)
                files.append(p)
    ext = [dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), dict(kind="gru", hidden=48, input_size=3, seed=483, n_rnn=3),
           dict(kind="lstm", hidden=16, input_size=2, seed=164, n_rnn=4, in_skip=1), dict(kind="gru", hidden=128, input_size=3, seed=128),
           dict(kind="lstm", hidden=112, input_size=1, seed=112, in_skip=1), dict(kind="conv", hidden=16, input_size=1, seed=1608),
           dict(kind="lstm", hidden=36, input_size=1, seed=36, n_rnn=2)]
    for i, kw in enumerate(ext):
        p = str(tmp_path / f"ext{i}.json")
        modelgen.write_model(modelgen.make_model(**kw), p)
        files.append(p)
    bundled = os.path.join(ROOT, "tests", "golden", "models")
    files += [os.path.join(bundled, f) for f in sorted(os.listdir(bundled)) if f.endswith(".json")]
    out = _run(harness, files)
    loaded = int(out.split("asan_harness:")[1].split("loaded")[0])
    assert loaded == len(files), out


def test_mutated_model_files_under_sanitizers(harness, tmp_path):
    base = json.dumps(modelgen.make_model("gru", 8, 2, seed=3, in_skip=1, in_gain=-3.0)).encode()
    rs = np.random.RandomState(4321)
    alphabet = b'{}[],:"0123456789.-eE nulltruefalse\\\x00\xff'
    files = []
    for it in range(600):
        t = bytearray(base)
        kind = it % 5
        if kind == 0:
            t = t[:rs.randint(0, len(t))]
        elif kind == 1:
            a = rs.randint(0, len(t)); b = min(len(t), a + rs.randint(1, 40))
            del t[a:b]
        elif kind == 2:
            for _ in range(rs.randint(1, 6)):
                t[rs.randint(0, len(t))] = alphabet[rs.randint(0, len(alphabet))]
        elif kind == 3:
            a = rs.randint(0, len(t))
            t[a:a] = bytes(alphabet[rs.randint(0, len(alphabet))] for _ in range(rs.randint(1, 20)))
        else:
            a = rs.randint(0, len(t)); b = min(len(t), a + rs.randint(1, 200))
            t[a:a] = t[a:b]
        p = str(tmp_path / f"m{it}.json")
        open(p, "wb").write(bytes(t))
        files.append(p)
    for name, doc in (("empty", b""), ("deep", b"[" * 100000), ("nest", b'{"in_shape": [null, null, 1], "layers": ' + b"[" * 5000 + b"]" * 5000 + b"}"),
                      ("long", b'{"a": "' + b"x" * 1000000 + b'"}'), ("missing", None)):
        p = str(tmp_path / f"{name}.json")
        if doc is not None:
            open(p, "wb").write(doc)
        files.append(p)
    out = _run(harness, files)
    assert "rejected" in out
