"""The SHIP leg: the hook-free part of the GPU suite once more, on the bits that ship.

The suite loads aidadsp-lv2_amd/lib/hooks/libaidax_hip.so (-DAIDAX_TEST_HOOKS: kernel forms can be forced, faults injected);
bench.py, smoke(), the LV2 shell and the bundle load aidadsp-lv2_amd/lib/libaidax_hip.so, a second compilation of every kernel in
which the switches are constants. This test — collected LAST in a `-m gpu` session (tests/conftest.py) — starts a second pytest
session with AIDAX_SHIP_LEG=1: there tests/conftest.py points AIDAX_LIB at the shipped library (its child processes and the
in-process LV2 shell inherit it) and skips every test that asks for a hook. What remains — the reference's six goldens
(rt-neural-generic.cpp:900-955, bar TEST_MODEL_THR rt-neural-generic.h:182), the 54 variants in the form the pool picks, the chain
corners, ragged blocks and pre-run, the drift runs, cfg1 – cfg5 at full size, the conv stacks, the LV2 life cycle, the hub, the
soaks — must pass against the oracle there too, and every output array both legs drew from the library must be BIT-IDENTICAL
(same sources: a difference is a finding about the compiler or about a hook that leaks into the arithmetic)."""
import json
import os
import re
import subprocess
import sys

import pytest

from tests import conftest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tests whose outputs depend on wall-clock timing or on a random life drawn per process: they run on both legs, against the oracle,
# but their digests are not comparable from run to run (listed by what makes them so)
NOT_REPRODUCIBLE = (
    "test_two_full_size_stacked_pools_driven_at_once",      # two threads race for the device's chained-kernel gate: the number of blocks each serves differs from run to run
    "test_pools_of_one_process_on_several_threads",          # threads pick up work as they come: arrays per test differ from run to run
    "test_full_size_pool_processes_while_stacked_models_are_prepared_and_swapped_in",      # a model swap lands at whatever block the worker thread reaches it
    "test_worker_thread_prepares_while_the_audio_thread_processes",                         # the same
    "test_deadline_",                                        # the hub's deadline tests: which pass a block joins depends on the wall clock
)


@pytest.mark.gpu
def test_hook_free_suite_on_the_shipped_library(tmp_path):
    if conftest.SHIP_LEG:
        pytest.skip("this IS the ship leg")
    out = tmp_path / "ship.json"
    env = dict(os.environ, AIDAX_SHIP_LEG="1", AIDAX_DIGEST_OUT=str(out))
    env.pop("AIDAX_LIB", None)
    base = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-rs", "-rf"]
    r = subprocess.run(base + [os.path.join(ROOT, "tests")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    gp = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(gp):
        with open(os.path.join(gp, "ship_leg_pytest.txt"), "w") as f:
            f.write(r.stdout + r.stderr)
    tail = r.stdout[-6000:] + r.stderr[-2000:]
    # a test with a wall-clock bar (the hub's deadline tests, the latency audits) may miss it once on a busy box, on either leg: what
    # failed gets ONE more run on the shipped library, alone, and the report says so
    second_attempt = []
    if r.returncode != 0:
        failed = re.findall(r"^FAILED (\S+)", r.stdout, flags=re.M)
        assert failed and len(failed) <= 3, tail
        env2 = dict(env, AIDAX_DIGEST_OUT=str(tmp_path / "ship_retry.json"))
        r2 = subprocess.run(base + failed, cwd=ROOT, env=env2, capture_output=True, text=True, timeout=900)
        assert r2.returncode == 0, (failed, r2.stdout[-4000:])
        second_attempt = failed
    m = re.search(r"(\d+) passed", r.stdout)
    passed = int(m.group(1)) if m else 0
    m = re.search(r"(\d+) skipped", r.stdout)
    skipped = int(m.group(1)) if m else 0
    ship = json.loads(out.read_text())
    assert os.path.samefile(ship["lib"], conftest.SHIP_LIB)
    assert passed >= 120, (passed, skipped, tail)                      # the hook-free part is most of the suite

    # bit identity of the two builds on everything both legs ran
    mine = {k: [hex(v[0]), v[1]] for k, v in conftest.DIGESTS.items()}
    common = sorted(k for k in ship["digests"] if k in mine and k not in second_attempt and not any(t in k for t in NOT_REPRODUCIBLE))
    differ = [k for k in common if ship["digests"][k] != mine[k]]
    report = {"ship_passed": passed, "ship_skipped_need_a_hook": skipped, "tests_compared": len(common),
              "arrays_compared": sum(mine[k][1] for k in common), "differ": differ, "passed_on_second_attempt": second_attempt}
    if os.path.isdir(gp):
        with open(os.path.join(gp, "ship_leg.json"), "w") as f:
            json.dump(dict(report, ran_on_the_shipped_library=sorted(ship["digests"]), skipped_for_a_hook=ship.get("skipped_for_a_hook")), f, indent=1)
    print("ship leg:", report)
    assert len(common) >= 100, report
    assert not differ, report
