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
os.environ.setdefault("AIDAX_LIB", HOOKS_LIB)

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
    """measured parity errors of this session (tests/errlog.py) -> gpurun_out/parity_errors.json"""
    try:
        from tests import errlog
        out = os.path.join(ROOT, "gpurun_out")
        if errlog.LOG and os.path.isdir(out):
            import json
            with open(os.path.join(out, "parity_errors.json"), "w") as f:
                json.dump(dict(sorted(errlog.LOG.items())), f, indent=1)
    except Exception:
        pass
