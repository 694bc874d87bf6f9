"""cfg5's second operating point (round-4 review: the driver's box timed k_mfma_ls at 979 us against 697-714 here).

One HIP event pair PER LAUNCH of a bench workload's pool, in fresh child processes per variant (the library reads its switches
at load), with the placement stamps of AIDAX_TUNE bit 4096 (XCC_ID / HW_ID / 100 MHz clock at start and end per workgroup
of the LAST launch):

    python scratch/r05_modes.py                  # every variant, on this box
    python scratch/r05_modes.py child cfg5 400 tag [pre]     # one variant (what the parent starts)

Variants: cooperative / plain launch, a group's layers on one XCD (ids 8 apart) / on adjacent ids (AIDAX_TUNE bit 2), alone /
after the other workloads' regions in the same process (what bench.py does) / after a burst of host threads."""
import importlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AIDAX_LIB", os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "hooks", "libaidax_hip.so"))      # the build with the switches


def box_id():
    out = []
    for cmd in (["rocm-smi", "--showserial", "--showuniqueid", "--csv"], ["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"],
                ["rocm-smi", "--showmemorypartition", "--showcomputepartition", "--csv"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
            out.append(r.stdout.strip())
        except Exception as e:
            out.append(f"{cmd[0]}: {e}")
    return "\n".join(out)


def child(name, steps, tag, pre):
    import numpy as np
    import torch
    import bench
    ax = importlib.import_module("aidadsp-lv2_amd")
    W = ax.workloads
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)

    def region(nm, n_steps, per_launch):
        wl = bench.WORKLOADS[nm]
        path, _ = bench.workload_model_path(W, nm)
        S = wl["streams"]
        pool = ax.Pool(S, 256, 48000.0, device=0)
        pool.set_model(ax.Model(path), ax.START_WARMUP)
        pool.set_controls(ax.default_controls(**wl["controls"]))
        d_in = [torch.from_numpy(W.signal(S, 256, seed=0xA1DA + 7919 * r)).cuda() for r in range(8)]
        d_out = [torch.empty_like(t) for t in d_in]
        t0 = time.perf_counter()
        i = 0
        while time.perf_counter() - t0 < 0.15:
            for _ in range(16):
                pool.process_device(d_in[i % 8].data_ptr(), d_out[i % 8].data_ptr(), 256, st.cuda_stream)
                i += 1
            st.synchronize()
        if not per_launch:
            for i in range(n_steps):
                pool.process_device(d_in[i % 8].data_ptr(), d_out[i % 8].data_ptr(), 256, st.cuda_stream)
            torch.cuda.synchronize()
            pool.close()
            return None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
        ev[0].record(st)
        for i in range(n_steps):
            pool.process_device(d_in[i % 8].data_ptr(), d_out[i % 8].data_ptr(), 256, st.cuda_stream)
            ev[i + 1].record(st)
        torch.cuda.synchronize()
        t = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n_steps)])
        k = pool.kernel_name
        try:
            pool.sync()
            fault = "none"
        except Exception as e:
            fault = str(e)[:80]
        pool.close()
        return t, k, fault

    if pre == "others":
        for nm, n in (("cfg2", 2000), ("cfg3", 600), ("cfg4", 2000)):
            region(nm, n, False)
    elif pre == "hog":
        import threading
        stop = time.time() + 3.0
        def spin():
            x = 0
            while time.time() < stop:
                x += 1
        th = [threading.Thread(target=spin) for _ in range(4)]
        hogs = [subprocess.Popen([sys.executable, "-c", "import time\nt=time.time()\nwhile time.time()-t<3: pass"]) for _ in range(16)]
        for h in hogs:
            h.wait()
    trace = os.path.join(tempfile.mkdtemp(), "stamps.bin")
    os.environ["AIDAX_LP_TRACE_FILE"] = trace
    t, k, fault = region(name, steps, True)
    p50 = float(np.percentile(t, 50))
    slow = np.nonzero(t > 1.2 * p50)[0]
    print(f"{tag:34s} {k:12s} n={len(t)} min {t.min():7.1f} p50 {p50:7.1f} p95 {np.percentile(t, 95):7.1f} max {t.max():7.1f} us"
          f"  over 1.2 x p50: {len(slow)} {list(slow[:12])}  fault: {fault}", flush=True)
    if os.path.exists(trace) and (int(os.environ.get("AIDAX_TUNE", "0")) & 4096):
        w = np.fromfile(trace, dtype=np.uint32)
        nb = 256 if name == "cfg5" else 0
        if nb:
            rec = w[:3 * nb].reshape(nb, 3)
            xcc = rec[:, 0] & 15
            hw = rec[:, 0] >> 4
            se, cu = (hw >> 13) & 7, (hw >> 8) & 15
            adjacent = (int(os.environ.get("AIDAX_TUNE", "0")) & 2) != 0
            grp = np.array([b // 2 if adjacent else (b // 16) * 8 + (b & 7) for b in range(nb)])
            lay = np.array([b % 2 if adjacent else (b // 8) % 2 for b in range(nb)])
            t0 = rec[:, 1].astype(np.int64)
            t1 = rec[:, 2].astype(np.int64)
            base = t0.min()
            dur = (t1 - t0) / 100.0
            same = sum(1 for g in range(128) if len(set(xcc[grp == g])) == 1)
            per_xcc = np.bincount(xcc, minlength=8)
            ident = bool((xcc == (np.arange(nb) % 8)).all())
            cus = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
            print(f"    last launch: XCC == id % 8: {ident}; workgroups per XCC {per_xcc.tolist()}; distinct (xcc, se, cu) {cus}; groups with both layers on one XCC: {same}/128")
            print(f"    start spread {(t0.max() - base) / 100.0:.1f} us; duration per workgroup: layer 0 min {dur[lay == 0].min():.1f} p50 {np.median(dur[lay == 0]):.1f} max {dur[lay == 0].max():.1f}"
                  f" | layer 1 min {dur[lay == 1].min():.1f} p50 {np.median(dur[lay == 1]):.1f} max {dur[lay == 1].max():.1f} us; launch {(t1.max() - base) / 100.0:.1f} us")
            for x in range(8):
                m = xcc == x
                if m.any():
                    print(f"      xcc {x}: {int(m.sum()):3d} wgs, end of the last one {(t1[m].max() - base) / 100.0:7.1f} us, median duration {np.median(dur[m]):7.1f}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "none")
    print(box_id(), flush=True)
    variants = [
        ("coop, ids 8 apart", dict(AIDAX_LP_COOP="1", AIDAX_TUNE="4096"), "none"),
        ("plain, ids 8 apart", dict(AIDAX_LP_COOP="0", AIDAX_TUNE="4096"), "none"),
        ("coop, adjacent ids", dict(AIDAX_LP_COOP="1", AIDAX_TUNE=str(4096 + 2)), "none"),
        ("plain, adjacent ids", dict(AIDAX_LP_COOP="0", AIDAX_TUNE=str(4096 + 2)), "none"),
        ("coop, after cfg2/3/4 regions", dict(AIDAX_LP_COOP="1", AIDAX_TUNE="4096"), "others"),
        ("coop, after host burst", dict(AIDAX_LP_COOP="1", AIDAX_TUNE="4096"), "hog"),
        ("coop, ids 8 apart (again)", dict(AIDAX_LP_COOP="1", AIDAX_TUNE="4096"), "none"),
    ]
    steps = sys.argv[1] if len(sys.argv) > 1 else "400"
    for tag, env, pre in variants:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", "cfg5", steps, tag, pre], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        print(r.stdout.rstrip() or ("FAILED: " + r.stderr.strip()[-400:]), flush=True)


if __name__ == "__main__":
    main()
