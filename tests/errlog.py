"""Parity bars with a memory: every `bound(err, tol, tag)` asserts err < tol and remembers the largest error seen
per tag; at the end of a GPU session conftest.py writes them to gpurun_out/parity_errors.json. The tolerances in
the tests are set to ~10x those measured errors (a 100x regression fails), never above the reference's own 1e-5
(TEST_MODEL_THR, rt-neural-generic.h:182)."""
LOG = {}


def bound(err, tol, tag):
    err = float(err)
    LOG[tag] = max(LOG.get(tag, 0.0), err)
    assert err < tol, (tag, err, tol)
    return err
