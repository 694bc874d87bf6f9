"""Randomised soak (the reference has no such test; this is what a host does to a plugin): random
block sizes incl. 0 and 1, control flips on random streams, activate(), model swaps between six models (the two
extension families included) - three watched streams against the oracle's plugin mirror, in every launch form, at 70
streams (the resident forms) and at 4200 (the many-streams forms); and the hub under a host with a random life."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form", ["", "wave", "split", "quad", "mfma"])
def test_random_host_behaviour(form):
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    if form:
        from tests.conftest import needs_hook
        needs_hook("AIDAX_KERNEL")
        env["AIDAX_KERNEL"] = form                      # read at pool creation: one process per form
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak.py"), "250"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_host_behaviour_many_streams():
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    env["SOAK_STREAMS"] = "4200"                        # k_quad, k_nn, k_mfma, k_chain + k_conv_mfma
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak.py"), "200"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_host_behaviour_wide_models_many_streams():
    """... and with the wide one-layer cells and stacks as the models swapped between (SOAK_MODELS=wide: LSTM-64 / 80 / 40, GRU-80 / 64,
    LSTM-96 x 2) at 4200 streams: k_lstm_gs, k_mfma_ls1, k_gru_gs, k_mfma_ls in ranges of streams."""
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    env["SOAK_STREAMS"] = "4200"
    env["SOAK_MODELS"] = "wide"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak.py"), "200"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout and "k_lstm_gs" in r.stdout and "k_mfma_ls1" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_hub_host():
    """tests/soak_hub.py: attach / detach / skipped instances / block-size changes inside a period / model swaps of the
    hub, every delivered block against the instance's own oracle plugin one period late, pass count against the
    contract's."""
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_hub.py"), "300"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "hub soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_lv2_host():
    """tests/soak_lv2.py: the plugin shell under the mock host — patch:Set / restore requests overlapping in flight, late
    workers, late responses, failing loads, control moves, activate — against the oracle's plugin mirror."""
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_lv2.py"), "400"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "lv2 soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("model", ["lstm12", "stack"])
def test_hub_under_real_time_pacing_with_threads(model):
    """tests/soak_hub_rt.py: three host threads with jitter, dawdling and skipped periods, periods closed by the hub's
    deadline: an instance only ever gets the oracle's output of its previous block or (rarely) silence."""
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    env["SOAK_RT_MODEL"] = model
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_hub_rt.py"), "400"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "hub rt soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_abi_fuzz():
    """tests/fuzz_abi.py: random sequences of C-ABI calls with valid and cleanly-invalid arguments (null handles and
    buffers, streams / slots / sizes out of range, wrong modes, models a pool cannot host): a return code and a message,
    never a crash, and the pool still matches the oracle afterwards."""
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_abi.py"), "25"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "abi fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_lv2_host_in_hub_mode():
    """tests/soak_lv2_hub.py: four plugin instances of one process with AIDAX_HUB, moving between the hubs of four model
    files by patch:Set with late workers and responses, controls moving while a block still waits for its pass."""
    env = dict(os.environ)
    env.pop("AIDAX_KERNEL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak_lv2_hub.py"), "200"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "lv2 hub soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
