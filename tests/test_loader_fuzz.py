"""The model loader takes files from users: mutated / truncated json must come back as an error code
(or load as a valid model), never crash or hang. Deterministic mutations of a small valid file."""
import ctypes as C
import importlib
import json

import numpy as np

from tests import modelgen

ax = importlib.import_module("aidadsp-lv2_amd")


def _load(text: bytes) -> int:
    L = ax.lib()
    h = C.c_void_p()
    rc = L.aidax_model_load_memory(text, len(text), b"fuzz", C.byref(h))
    if h.value:
        assert rc == 0
        L.aidax_model_free(h)
    else:
        assert rc < 0 and L.aidax_last_error()
    return rc


def test_mutated_model_files_never_crash_the_loader():
    base = json.dumps(modelgen.make_model("gru", 8, 2, seed=3, in_skip=1, in_gain=-3.0)).encode()
    assert _load(base) == 0
    rs = np.random.RandomState(1234)
    alphabet = b'{}[],:"0123456789.-eE nulltruefalse\\\x00\xff'
    outcomes = {0: 0}
    for it in range(1500):
        t = bytearray(base)
        kind = it % 5
        if kind == 0:                                   # truncate
            t = t[:rs.randint(0, len(t))]
        elif kind == 1:                                 # delete a span
            a = rs.randint(0, len(t)); b = min(len(t), a + rs.randint(1, 40))
            del t[a:b]
        elif kind == 2:                                 # overwrite bytes
            for _ in range(rs.randint(1, 6)):
                t[rs.randint(0, len(t))] = alphabet[rs.randint(0, len(alphabet))]
        elif kind == 3:                                 # insert junk
            a = rs.randint(0, len(t))
            t[a:a] = bytes(alphabet[rs.randint(0, len(alphabet))] for _ in range(rs.randint(1, 20)))
        else:                                           # duplicate a span (nesting / repetition)
            a = rs.randint(0, len(t)); b = min(len(t), a + rs.randint(1, 200))
            t[a:a] = t[a:b]
        rc = _load(bytes(t))
        outcomes[rc] = outcomes.get(rc, 0) + 1
    assert set(outcomes) <= {0, -1, -2, -3, -4}
    assert outcomes.get(-3, 0) > 100                    # most mutations are json / shape errors


def test_pathological_documents():
    assert _load(b"") < 0
    assert _load(b"[" * 100000) < 0                     # deep nesting must not blow the stack
    assert _load(b'{"in_shape": [null, null, 1], "layers": ' + b"[" * 5000 + b"]" * 5000 + b"}") < 0
    assert _load(b'{"a": "' + b"x" * 1000000 + b'"}') < 0
    big = json.dumps({"in_shape": [None, None, 1], "layers": [{"type": "lstm", "shape": [None, None, 8],
                                                                "weights": [[1e308] * 10]}]}).encode()
    assert _load(big) < 0
