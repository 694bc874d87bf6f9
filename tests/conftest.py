import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite runs on the library built with -DAIDAX_TEST_HOOKS (lib/hooks/): it forces kernel forms, injects faults and reads measurement
# stamps through environment switches that the shipped library (lib/libaidax_hip.so: bench.py, smoke(), the LV2 shell, the bundle) does not
# have. Same sources, same kernels. (An explicit AIDAX_LIB — an A/B build — wins; subprocesses inherit it.)
SHIP_LIB = os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "libaidax_hip.so")
HOOKS_LIB = os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "hooks", "libaidax_hip.so")
# The SHIP leg (AIDAX_SHIP_LEG=1; started by tests/test_gpu_ship.py as a second session at the end of the first): the same suite on the
# library that bench.py, smoke(), the LV2 shell and the bundle load. A test that asks for a test hook (monkeypatch.setenv of a name the
# sources read through AIDAX_HOOK_ENV) is skipped there — the shipped library has no such switch — and every other test runs on the
# shipped bits against the same oracle. Both legs keep a digest of every output array a test drew from the library (`DIGESTS`, below):
# the two builds must agree bit for bit on what both ran.
SHIP_LEG = os.environ.get("AIDAX_SHIP_LEG") == "1"
if SHIP_LEG:
    os.environ["AIDAX_LIB"] = SHIP_LIB
else:
    os.environ.setdefault("AIDAX_LIB", HOOKS_LIB)


def hook_names():
    """the environment names only the test-hooks build reads (csrc: AIDAX_HOOK_ENV("..."))"""
    import glob
    import re
    names = set()
    for path in glob.glob(os.path.join(ROOT, "aidadsp-lv2_amd", "csrc", "*")):
        with open(path, errors="replace") as f:
            names |= set(re.findall(r'AIDAX_HOOK_ENV\("([A-Z0-9_]+)"\)', f.read()))
    return names


HOOK_NAMES = hook_names()


def needs_hook(name):
    """for tests that hand a hook to a child process through its environment (monkeypatch.setenv is covered by the fixture below)"""
    assert name in HOOK_NAMES, name
    if SHIP_LEG:
        pytest.skip(f"{name}: a test hook — the shipped library has none")

# test id -> order-independent digest (sum of blake2b over every output array, mod 2^128) and the number of arrays
DIGESTS = {}
SKIPPED_FOR_A_HOOK = {}       # ship leg: test id -> the hook it asked for
_current = [None]


def _note_output(*arrays):
    if _current[0] is None:
        return
    import hashlib
    import numpy as np
    acc, n = DIGESTS.get(_current[0], (0, 0))
    for a in arrays:
        a = np.ascontiguousarray(a)
        h = hashlib.blake2b(a.tobytes(), digest_size=16)
        h.update(str((a.dtype.str, a.shape)).encode())
        acc = (acc + int.from_bytes(h.digest(), "little")) % (1 << 128)
        n += 1
    DIGESTS[_current[0]] = (acc, n)


def _wrap_outputs():
    """Every array the binding (and the mock LV2 host) hands back to a test goes through _note_output. Wrapping happens here, in the
    test harness: aidadsp-lv2_amd/binding.py stays a marshaller."""
    import functools
    import importlib
    binding = importlib.import_module("aidadsp-lv2_amd.binding")
    from tests import lv2host

    def wrap(cls, name, pick=lambda r: (r,)):
        fn = getattr(cls, name)
        if getattr(fn, "_digested", False):
            return

        @functools.wraps(fn)
        def inner(*a, **k):
            r = fn(*a, **k)
            _note_output(*pick(r))
            return r
        inner._digested = True
        setattr(cls, name, inner)
    wrap(binding.Pool, "process")
    wrap(binding.Pool, "collect")
    wrap(binding.Pool, "read_state", lambda r: r)
    wrap(binding.Hub, "run")
    wrap(binding.Model, "self_test", lambda r: (r[2],))
    wrap(binding.Model, "forward")
    wrap(lv2host.Host, "run")


GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(GOLDEN, "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no binaries (they are git-ignored): build once, like __graft_entry__.build()
    lib = os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "libaidax_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "libaidax_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(HOOKS_LIB) and os.path.exists(orc)):
        import shutil
        import subprocess
        if shutil.which("hipcc") and shutil.which("make"):
            subprocess.run(["make", "-j8", "all"], cwd=ROOT, check=False, stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    """the ship leg's driver (tests/test_gpu_ship.py) runs last: it compares with what THIS session has produced"""
    last = [i for i in items if "test_gpu_ship.py" in i.nodeid]
    if last:
        items[:] = [i for i in items if i not in last] + last


@pytest.fixture(autouse=True)
def _leg_rules(request, monkeypatch):
    """Per test: outputs are digested under the test's id; on the ship leg a request for a test hook skips the test."""
    is_gpu = "gpu" in request.node.keywords
    if is_gpu:
        _wrap_outputs()
        _current[0] = request.node.nodeid
    if SHIP_LEG:
        plain = monkeypatch.setenv

        def setenv(name, value, prepend=None):
            if name in HOOK_NAMES:
                DIGESTS.pop(request.node.nodeid, None)
                SKIPPED_FOR_A_HOOK[request.node.nodeid] = name
                pytest.skip(f"{name}: a test hook — the shipped library has none")
            return plain(name, value, prepend)
        monkeypatch.setenv = setenv
    yield
    _current[0] = None


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """A failed GPU test may leave pools and hubs open (the traceback keeps them alive) — among them one that holds a device's
    right to the chained kernels (LpGate), after which every later stacked pool of the session would get k_mfma and fail on its
    kernel name: one failure would read as ten. Close what a failed test left behind."""
    outcome = yield
    rep = outcome.get_result()
    if rep.when == "call" and rep.failed and "gpu" in item.keywords:
        try:
            import importlib
            binding = importlib.import_module("aidadsp-lv2_amd.binding")
            for obj in list(binding.live_handles):
                try:
                    obj.close()
                except Exception:
                    pass
        except Exception:
            pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def bundled_models():
    return sorted(os.path.join(MODELS, f) for f in os.listdir(MODELS) if f.endswith(".json"))


def pytest_sessionfinish(session, exitstatus):
    """measured parity errors of this session (tests/errlog.py) -> gpurun_out/parity_errors.json; the ship leg's output digests and
    errors -> AIDAX_DIGEST_OUT (read back by tests/test_gpu_ship.py in the session that started it)"""
    if os.environ.get("AIDAX_DIGEST_OUT"):
        import json
        from tests import errlog
        with open(os.environ["AIDAX_DIGEST_OUT"], "w") as f:
            json.dump({"digests": {k: [hex(v[0]), v[1]] for k, v in DIGESTS.items()}, "errors": errlog.LOG, "skipped_for_a_hook": SKIPPED_FOR_A_HOOK,
                       "lib": os.environ.get("AIDAX_LIB")}, f, indent=1)
    if SHIP_LEG:
        return
    try:
        from tests import errlog
        out = os.path.join(ROOT, "gpurun_out")
        if errlog.LOG and os.path.isdir(out):
            import json
            with open(os.path.join(out, "parity_errors.json"), "w") as f:
                json.dump(dict(sorted(errlog.LOG.items())), f, indent=1)
    except Exception:
        pass
