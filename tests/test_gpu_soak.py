"""Randomised soak (the reference has no such test; this is what a host does to a plugin): random
block sizes incl. 0 and 1, control flips on random streams, activate(), model swaps between four models -
three watched streams of 70 against the oracle's plugin mirror, in every launch form."""
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
        env["AIDAX_KERNEL"] = form                      # read at pool creation: one process per form
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak.py"), "250"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
