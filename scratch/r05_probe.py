"""A cheap probe for cfg5's slow operating point, without torch (a fresh box pays 1-2 minutes for the first `import torch`):
per-launch HIP event pairs through ctypes, the box's identity, and its clocks / power WHILE the kernel runs.

    python scratch/r05_probe.py [steps] [workload]        # on the GPU box: ~25 s
    python scratch/r05_probe.py child <workload> <steps> <tag>

If the probe finds the slow point (p50 over 850 us on cfg5) it runs the variants of r05_modes.py in the same call."""
import ctypes as C
import importlib
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AIDAX_LIB", os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "hooks", "libaidax_hip.so"))      # the build with the switches
os.environ.setdefault("AIDAX_NO_TORCH", "1")

import numpy as np  # noqa: E402

WL = {
    "cfg2": dict(model=dict(kind="lstm", hidden=32, input_size=1, seed=32), streams=1024),
    "cfg3": dict(model=dict(kind="gru", hidden=64, input_size=3, seed=64), streams=4096),
    "cfg4": dict(model=dict(kind="conv", hidden=16, input_size=1, seed=1608), streams=1024),
    "cfg5": dict(model=dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), streams=2048),
}


def smi(*args):
    try:
        return subprocess.run(["rocm-smi", *args, "--csv"], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e:
        return f"rocm-smi: {e}"


class Hip:
    def __init__(self):
        self.l = C.CDLL("libamdhip64.so")

    def ok(self, rc):
        if rc != 0:
            raise RuntimeError(f"hip error {rc}")

    def malloc(self, n):
        p = C.c_void_p()
        self.ok(self.l.hipMalloc(C.byref(p), C.c_size_t(n)))
        return p

    def h2d(self, dst, arr):
        self.ok(self.l.hipMemcpy(dst, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1))

    def stream(self):
        s = C.c_void_p()
        self.ok(self.l.hipStreamCreate(C.byref(s)))
        return s

    def event(self):
        e = C.c_void_p()
        self.ok(self.l.hipEventCreate(C.byref(e)))
        return e

    def record(self, e, s):
        self.ok(self.l.hipEventRecord(e, s))

    def sync(self):
        self.ok(self.l.hipDeviceSynchronize())

    def elapsed_us(self, a, b):
        ms = C.c_float()
        self.ok(self.l.hipEventElapsedTime(C.byref(ms), a, b))
        return ms.value * 1e3


def child(name, steps, tag, watch):
    ax = importlib.import_module("aidadsp-lv2_amd")
    W = ax.workloads
    hip = Hip()
    wl = WL[name]
    S = wl["streams"]
    path = W.write_model(W.make_model(**wl["model"]), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, 256, 48000.0, device=0)
    pool.set_model(ax.Model(path), ax.START_WARMUP)
    st = hip.stream()
    x = W.signal(S, 256, seed=0xA1DA)
    d_in, d_out = hip.malloc(x.nbytes), hip.malloc(x.nbytes)
    hip.h2d(d_in, x)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(16):
            pool.process_device(d_in.value, d_out.value, 256, st.value)
        hip.sync()
    seen = []
    if watch:
        def sample():
            time.sleep(0.15)
            seen.append(smi("--showclocks", "--showpower", "--showtemp"))
        th = threading.Thread(target=sample)
        th.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.5:            # keep the kernel running while rocm-smi reads the clocks
            for _ in range(16):
                pool.process_device(d_in.value, d_out.value, 256, st.value)
            hip.sync()
        th.join()
    ev = [hip.event() for _ in range(steps + 1)]
    hip.record(ev[0], st)
    for i in range(steps):
        pool.process_device(d_in.value, d_out.value, 256, st.value)
        hip.record(ev[i + 1], st)
    hip.sync()
    t = np.array([hip.elapsed_us(ev[i], ev[i + 1]) for i in range(steps)])
    p50 = float(np.percentile(t, 50))
    slow = np.nonzero(t > 1.2 * p50)[0]
    print(f"{tag:30s} {pool.kernel_name:14s} n={len(t)} min {t.min():7.1f} p50 {p50:7.1f} p95 {np.percentile(t, 95):7.1f} max {t.max():7.1f} us"
          f"  over 1.2 x p50: {len(slow)} {list(slow[:10])}", flush=True)
    for s in seen:
        print("    while running: " + s.replace("\n", "\n                   "), flush=True)
    pool.close()
    return p50


def long_run(name, seconds):
    """the kernel back to back for `seconds`, every launch timed, rocm-smi every two seconds: does the board's power management
    find another operating point under a sustained load?"""
    ax = importlib.import_module("aidadsp-lv2_amd")
    W = ax.workloads
    hip = Hip()
    wl = WL[name]
    S = wl["streams"]
    path = W.write_model(W.make_model(**wl["model"]), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, 256, 48000.0, device=0)
    pool.set_model(ax.Model(path), ax.START_WARMUP)
    st = hip.stream()
    x = W.signal(S, 256, seed=0xA1DA)
    d_in, d_out = hip.malloc(x.nbytes), hip.malloc(x.nbytes)
    hip.h2d(d_in, x)
    stop = [False]
    smis = []

    def watch():
        while not stop[0]:
            smis.append((time.perf_counter(), smi("--showclocks", "--showpower", "--showtemp").splitlines()[-1]))
            time.sleep(1.5)
    th = threading.Thread(target=watch)
    th.start()
    B = 64
    ev = [hip.event() for _ in range(B + 1)]
    ts, when = [], []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        hip.record(ev[0], st)
        for i in range(B):
            pool.process_device(d_in.value, d_out.value, 256, st.value)
            hip.record(ev[i + 1], st)
        hip.sync()
        now = time.perf_counter() - t0
        for i in range(B):
            ts.append(hip.elapsed_us(ev[i], ev[i + 1]))
            when.append(now)
    stop[0] = True
    th.join()
    t = np.array(ts)
    when = np.array(when)
    p50 = float(np.percentile(t, 50))
    print(f"{name} for {seconds:.0f} s: n={len(t)} min {t.min():.1f} p50 {p50:.1f} p95 {np.percentile(t, 95):.1f} p99.9 {np.percentile(t, 99.9):.1f} max {t.max():.1f} us; over 1.2 x p50: {int((t > 1.2 * p50).sum())}")
    for a in range(0, int(seconds), 2):
        m = (when >= a) & (when < a + 2)
        if m.any():
            print(f"   t = {a:3d}..{a + 2:3d} s: p50 {np.percentile(t[m], 50):7.1f} max {t[m].max():7.1f} us   n over 850 us: {int((t[m] > 850).sum())}")
    for w, line in smis:
        print(f"   smi at {w - t0:6.1f} s: {line}")
    pool.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        print(smi("--showserial", "--showuniqueid"))
        print(smi("--showmaxpower", "--showperflevel", "--showmemorypartition", "--showcomputepartition"))
        print(smi("--showclocks", "--showpower", "--showtemp").splitlines()[0])
        return long_run(sys.argv[2] if len(sys.argv) > 2 else "cfg5", float(sys.argv[3]) if len(sys.argv) > 3 else 30.0)
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4], len(sys.argv) > 5 and sys.argv[5] == "watch")
        return
    steps = sys.argv[1] if len(sys.argv) > 1 else "300"
    print(smi("--showserial", "--showuniqueid"))
    print(smi("--showclocks", "--showpower", "--showtemp", "--showmemorypartition", "--showcomputepartition", "--showperflevel"), flush=True)

    def run(name, tag, env=None, watch=True):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", name, steps, tag] + (["watch"] if watch else []),
                           env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=300)
        print(r.stdout.rstrip() or ("FAILED: " + r.stderr.strip()[-400:]), flush=True)
        try:
            return float(r.stdout.split("p50")[1].split()[0])
        except Exception:
            return 0.0
    p50 = run("cfg5", "cfg5 coop")
    run("cfg3", "cfg3", watch=False)
    if p50 > 850.0:
        print("SLOW POINT FOUND: variants", flush=True)
        run("cfg5", "cfg5 plain", dict(AIDAX_LP_COOP="0"))
        run("cfg5", "cfg5 coop adjacent ids", dict(AIDAX_TUNE="2"))
        run("cfg5", "cfg5 fp32 MFMA (k_mfma_lp)", dict(AIDAX_LP_SPLIT="0"))
        run("cfg5", "cfg5 k_mfma (no ring)", dict(AIDAX_KERNEL="mfma"))
        run("cfg2", "cfg2", watch=False)
        run("cfg4", "cfg4", watch=False)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "r05_modes.py"), "child", "cfg5", "200", "stamps (coop, 8 apart)"],
                           env={k: v for k, v in dict(os.environ, AIDAX_TUNE="4096").items() if k != "AIDAX_NO_TORCH"}, capture_output=True, text=True, timeout=400)
        print(r.stdout.rstrip() or r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
