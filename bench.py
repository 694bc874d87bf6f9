#!/usr/bin/env python3
"""bench.py — headline benchmark of the rt-neural-generic hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU. Either the caller starts the ranks (python -m torch.distributed.run --nnodes=1
--nproc-per-node N ... bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or — when they
are absent — this process starts them itself, as children, BEFORE it touches HIP or imports torch, waits for
them and relays rank 0's json line as its own last line of stdout.

Workload (BASELINE.json configs[1], "cfg2"): LSTM hidden=32 amp model (synthetic Glorot-style weights, seed 32 —
no checkpoints offline), 1024 concurrent 48 kHz mono streams PER GPU, 256-frame blocks, plugin controls at their
TTL defaults (LPF 66.216 %, DC blocker on, EQ post flat, gains 0 dB): the full run() chain of
rt-neural-generic.cpp:621-659. One "step" = one aidax_pool_process_device pass over one [1024][256] fp32 block
already resident in HBM.

Streams are independent, so N GPUs run N x 1024 streams with no data-path collective ("weak" scaling); RCCL
carries one MAX(elapsed) / SUM(samples) reduction at the end.

Prints ONE json line (rank 0) with the driver's contract keys plus
  roofline          HBM roofline of the pass from ALGORITHMIC bytes (8 B per sample per stream: 4 in + 4 out,
                    SURVEY §8(d)) over the average launch duration measured with HIP events on the launch stream
  roofline_compute  the same pass against the fp32 peak from ALGORITHMIC flops — the roofline
                    that actually binds this path (1064 flop/B against a machine balance of ~20)
  other_workloads   short timed regions of the other BASELINE configs at their per-GPU sizes in the same process,
                    each with its rooflines, max-abs error and its own cpu_baseline (N = 1: cfg3, cfg4, cfg5; N > 1:
                    cfg4 and cfg5 — BASELINE's 8-GPU workloads — on every rank, reduced like the headline)
  ranks             world size, collective backend and every rank's own elapsed time of the headline region
  realtime_case     the LV2 case of SURVEY §8(d): ONE stream, 256-frame blocks — wall time per
                    aidax_pool_process call (pinned staging + launch + wait) next to the CPU oracle on one thread
  realtime_paced    the same call made the way a host makes it — ONE call per audio period (1 333 / 2 667 / 5 333 us = 64 / 128 / 256
                    frames at 48 kHz; the caller spins until the deadline, the GPU idles in between): the one-instance LV2 pool, a
                    1024-seat hub (one aidax_hub_run per seat and period) and a cfg2-size pool; p50 / p99 / p99.9 / max beside the
                    back-to-back figures, the kernel's own duration (HIP events) and the shader clock read while the paced loop runs
  host_inclusive    cfg2 blocks from and to HOST memory (what the reference's run() is handed: rt-neural-generic.cpp:484-487):
                    aidax_pool_process (pageable), submit / collect with one block in flight, submit_to with registered buffers —
                    us per block and samples/s, never `value`
  cpu_baseline      the CPU oracle (a port, not RTNeural) timed on this host's cores on a bounded sample of
                    the same workload (N=1, rank 0 only)
  max_abs_err       of the TIMED pool's first blocks (sixteen streams spread over it) against the CPU oracle, with the kernel that
                    produced it (max_abs_err_kernel, parity{...}); every other_workloads entry carries its own
  per_launch_us     one HIP event pair per launch after the timed region: min / p50 / p95 / max, launches over 1.2 x p50
  gpu_state_while_running   shader clock / socket power read by rocm-smi while the workload's kernel runs
  roofline.traffic  HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: two child runs of this
                    script under the profiler, N = 1 only); when the profiler is not usable, the committed profile of
                    the same kernel sources (hash-checked), else null
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FRAMES = 256
RING = 8                  # distinct input blocks cycled through HBM
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector FP32 peak = v_mfma_f32_16x16x4_f32 peak (64 FLOP/clk/SIMD)
BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak (~2.5 PFLOP/s)
SPLIT_PRODUCTS = 6        # k_gru_gs / k_mfma_ls / k_conv_ms / k_conv_st: bf16 term products issued per fp32 product (operands split exactly into three bf16 terms)
N_SIMDS = 1024            # 256 CUs x 4
NOMINAL_GHZ = 2.4
ALGO_BYTES_PER_SAMPLE = 8
PREROLL_S = 0.35          # un-timed passes before the warm-up: clocks ramp, caches and TLBs fill

# The headline line is cfg2 (BASELINE.json configs[1]). The other GPU configs run as short regions after it
# (other_workloads) or on their own with --workload; they are parity-test cases, not the line the driver records.
WORKLOADS = {
    "cfg2": dict(model=dict(kind="lstm", hidden=32, input_size=1, seed=32), streams=1024, controls={},
                 flops=8512 + 63 + 6, bound="hbm",
                 text="cfg2: LSTM-32 amp model, 1024 streams/GPU x 256-frame blocks, full run() chain, TTL-default controls"),
    "cfg3": dict(model=dict(kind="gru", hidden=64, input_size=3, seed=64), streams=4096,
                 controls=dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0,
                               presence_boost_db=3.0, param1=0.5, param2=0.3),
                 flops=2 * (3 * 64 * (64 + 3) + 64) + 63 + 6, bound="hbm", split_flops=2 * 3 * 64 * 64, fetch_wide="audio",
                 text="cfg3: GRU-64 conditioned (PARAM1+PARAM2) pedal model + 5-band EQ post, 4096 streams x 256-frame blocks"),
    "cfg4": dict(model=dict(kind="conv", hidden=16, input_size=1, seed=1608), streams=1024, controls={},
                 flops=2 * (3 * 1 * 16 + 7 * 3 * 16 * 16 + 16) + 63 + 6, bound="hbm", split_flops=2 * 7 * 3 * 16 * 16, fetch_wide="all",
                 text="cfg4: dilated conv1d stack (8 layers, 16 channels, k=3), 1024 streams/GPU (8192 over 8 GPUs) x 256-frame blocks"),
    "cfg5": dict(model=dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), streams=2048, controls={},
                 flops=2 * (384 * (1 + 96) + 384 * (96 + 96) + 96), bound="mfma", split_flops=2 * (384 * 96 + 384 * 192), fetch_wide="all",
                 text="cfg5: LSTM-96 x2 model, 2048 streams/GPU (16384 over 8 GPUs) x 256-frame blocks, matrix-core kernel"),
}
OTHER_STEPS = {"cfg3": 600, "cfg4": 2000, "cfg5": 120}      # ~0.25 s of GPU time each
DIST_STEPS = {"cfg2": 1000, "cfg3": 400, "cfg4": 1000, "cfg5": 300}    # launches timed one by one after a workload's timed region


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def reduce_results(elapsed_s: float, samples: float, world: int, backend_device="cuda", force=False):
    """MAX(elapsed) and SUM(samples) over ranks — the only collective of the path.
    force: go through the process group even at world size 1 (--force-dist: the RCCL branch on a one-GPU box)."""
    if world == 1 and not force:
        return elapsed_s, samples
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=backend_device)
    n = torch.tensor([samples], dtype=torch.float64, device=backend_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), float(n.item())


def stream_range(rank: int, world: int, total_streams: int):
    """Contiguous stream ranges per rank."""
    lo = total_streams * rank // world
    hi = total_streams * (rank + 1) // world
    return lo, hi


def launch_ranks(n_gpus: int, argv):
    """No launcher around us: start one rank per GPU as child processes (this process has not touched HIP and
    never will), wait, relay rank 0's json line. Returns the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)              # rank chatter (RCCL banner, launcher notes) stays off stdout
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n_gpus}-rank run failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def workload_model_path(W, name: str):
    j = W.make_model(**WORKLOADS[name]["model"])
    d = tempfile.mkdtemp(prefix="aidax_bench_")
    return W.write_model(j, os.path.join(d, f"{name}.json")), j


def usable_cores() -> int:
    """Cores this process may really use: affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def kernel_sources_sha16() -> str:
    """Hash of everything the device code is built from (csrc/*.hip, csrc/*.h, the Makefile's flags): a committed
    counter profile describes the benchmarked kernel only if it was taken on the same sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "aidadsp-lv2_amd", "csrc")
    for fn in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + [os.path.join(ROOT, "Makefile")]):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def lib_sha16(ax) -> str:
    import hashlib
    return hashlib.sha256(open(ax.lib_path(), "rb").read()).hexdigest()[:16]


def _kernel_matches(kernel_name: str, text: str) -> bool:
    """`k_lstm_pipe<32>` against a (possibly namespaced, templated) kernel name of a profile"""
    if "<" not in kernel_name:
        return kernel_name in text
    import re
    base, arg = kernel_name.split("<", 1)
    # (the instantiation may carry further template arguments behind the ones the name quotes: k_lstm_pipe4<32> is k_lstm_pipe4<32, false> there)
    return re.search(re.escape(base) + r"<\s*" + re.escape(arg.rstrip(">").strip()) + r"\s*[,>]", text) is not None


def measure_traffic_live(workload: str, kernel_name: str, wide_read_bytes: float, timeout_s: float = 100.0):
    """HBM bytes per launch of the benchmarked kernel, measured now: two short child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are collected without any trace
    domain). Units and the gfx950 correction as MI355X_MICROARCH.md prescribes: both counters are KB; FETCH_SIZE
    reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane) and narrower reads at face value,
    so the pass's one wide read — the audio block, `wide_read_bytes` — is added back once. None when the profiler
    cannot be run here."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3")
    if not prof:
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None                                # this run is being profiled itself: no profiler under a profiler
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="aidax_pmc_")
        cmd = [prof, "--output-format", "csv", "--pmc", ctr, "-d", d, "-o", "r", "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload, "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-check", "--no-others", "--no-traffic", "--no-dist"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        except Exception:
            return None
        if r.returncode != 0:
            return None
        got = []
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if row.get("Counter_Name") == ctr and _kernel_matches(kernel_name, row.get("Kernel_Name", "")):
                    got.append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if len(got) < 20:
            return None
        got.sort()
        vals[ctr] = got[len(got) // 2]
    return {"bytes": vals["FETCH_SIZE"] * 1024.0 + 0.5 * wide_read_bytes + vals["WRITE_SIZE"] * 1024.0,
            "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this command (200 steps each), median per launch"}


def pmc_traffic_bytes(kernel_name: str, wide_read_bytes: float):
    """Fallback: the same figure from the COMMITTED rocprofv3 PMC passes (profiles/*_pmc_summary.txt, latest round
    that has this kernel) — only if that profile was taken on the kernel sources this library was built from
    (`kernel_src_sha16=` line of the summary); a profile of other sources says nothing about this build. None then."""
    import glob
    import re
    best = None
    mine = kernel_sources_sha16()
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt"))):
        fetch = write = None
        text = open(fn).read()
        m = re.search(r"kernel_src_sha16=([0-9a-f]{16})", text)
        if not m or m.group(1) != mine:
            continue
        for line in text.splitlines():
            if not _kernel_matches(kernel_name, line):
                continue
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ median=([0-9.e+]+)", line)
            if m and m.group(1) == "FETCH_SIZE":
                fetch = float(m.group(2))
            elif m:
                write = float(m.group(2))
        if fetch is not None and write is not None:
            best = {"bytes": fetch * 1024.0 + 0.5 * wide_read_bytes + write * 1024.0,
                    "fetch_size_kb": fetch, "write_size_kb": write, "source": os.path.basename(fn)}
    return best


def profile_counters(kernel_name: str):
    """Median per-launch counters of `kernel_name` from the committed rocprofv3 PMC summaries taken on THESE kernel
    sources (hash-checked like pmc_traffic_bytes): {counter: median}; {} when there is no such profile."""
    import glob
    import re
    mine = kernel_sources_sha16()
    out = {}
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt"))):
        text = open(fn).read()
        m = re.search(r"kernel_src_sha16=([0-9a-f]{16})", text)
        if not m or m.group(1) != mine:
            continue
        for line in text.splitlines():
            if not _kernel_matches(kernel_name, line):
                continue
            m = re.search(r"\b([A-Z][A-Z0-9_]+)\s+n=\s*\d+ median=([0-9.e+]+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
                out["_source"] = os.path.basename(fn)
    return out


def add_profile_figures(hbm: dict, comp: dict, name: str, S: int, kernel: str, kernel_ms: float):
    """HBM traffic and matrix-pipe occupancy of a workload's kernel from the committed same-source profile, into its rooflines."""
    c = profile_counters(kernel.split("+")[-1])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        audio = 4.0 * S * N_FRAMES
        wide = WORKLOADS[name].get("fetch_wide", "audio")
        fetch = c["FETCH_SIZE"] * 1024.0
        # FETCH_SIZE reports half of a 16 B/lane streaming read (MI355X_MICROARCH.md): the audio block always is one; cfg4's layer
        # history and cfg5's hand-over ring are read that way too ("all": 2 x FETCH)
        hbm["traffic"] = (2.0 * fetch if wide == "all" else fetch + 0.5 * audio) + c["WRITE_SIZE"] * 1024.0
        hbm["traffic_source"] = f"committed profile of the same kernel sources: {c['_source']} (FETCH_SIZE {c['FETCH_SIZE']:.0f} KB, WRITE_SIZE {c['WRITE_SIZE']:.0f} KB per launch)"
        comp["traffic"] = hbm["traffic"]                      # (HBM bytes per launch: the same figure in either roofline object)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        comp["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMDS * kernel_ms * 1e-3 * NOMINAL_GHZ * 1e9)
        comp["mfma_busy_source"] = f"SQ_VALU_MFMA_BUSY_CYCLES {c['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} per launch / ({N_SIMDS} SIMDs x kernel time x {NOMINAL_GHZ} GHz), {c['_source']}"
    if "SQ_INSTS_MFMA" in c:
        comp["mfma_insts_per_launch"] = c["SQ_INSTS_MFMA"]


def cpu_baseline(W, j, name: str = "cfg2", target_s: float = 12.0):
    """The CPU oracle on all host cores over a bounded sample of the workload: up to 1024 of its streams (fewer for
    the heavy models, a multiple of the core count) over as many 256-frame blocks as fit `target_s`."""
    from oracle import oracle as O
    spec = O.parse_model(j)
    cores = usable_cores()
    c = O.default_controls(**WORKLOADS[name]["controls"])
    probe = 4 * cores
    secs, _ = O.cpu_bench(spec, c, W.signal(probe, N_FRAMES), n_blocks=1, warm_blocks=1, n_threads=cores, fast=True)
    per_stream_block = max(secs, 1e-6) / probe
    streams = 1024                            # the sample is bounded in blocks ...
    while streams > probe and 4 * streams * per_stream_block > target_s:
        streams //= 2                         # ... and, where four blocks of 1024 streams would not fit, in streams
    x = W.signal(streams, N_FRAMES)
    blocks = int(max(4, min(20000, target_s / (streams * per_stream_block))))
    secs, _ = O.cpu_bench(spec, c, x, n_blocks=blocks, warm_blocks=1, n_threads=cores, fast=True)
    sps = streams * N_FRAMES * blocks / secs
    return {"value": sps, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{streams} {name} streams x {blocks} blocks of {N_FRAMES} frames, "
                      f"full run() chain, C oracle with vectorised exp/tanh (-O3 -march=x86-64-v3, AVX2), {cores} pthreads, {secs:.1f} s"}


def realtime_case(ax, W, local: int):
    """SURVEY §8(d) CPU baseline (1), the LV2 real-time case: ONE stream, 256-frame blocks, for the bundled
    LSTM-12 (the TTL default model) and a synthetic LSTM-16 (BASELINE cfg1). GPU: wall time of one
    aidax_pool_process call on a one-stream pool — what the plugin's run() costs (pinned staging, launch, wait
    for the stream). CPU: the oracle on one thread."""
    import numpy as np
    from oracle import oracle as O
    out = []
    bundled = os.path.join(ROOT, "tests", "golden", "models", "tw40_california_clean_deerinkstudios.json")
    d = tempfile.mkdtemp(prefix="aidax_rt_")
    syn = W.write_model(W.make_model("lstm", 16, 1, seed=16), os.path.join(d, "lstm16.json"))
    x = W.signal(1, N_FRAMES, seed=5)
    for label, path in (("bundled LSTM-12 (tw40_california_clean)", bundled), ("synthetic LSTM-16", syn)):
        pool = ax.Pool(1, 8192, 48000.0, device=local)        # the pool the LV2 shell creates
        pool.set_model(ax.Model(path))
        for _ in range(300):
            pool.process(x)
        t = np.empty(3000)
        for i in range(t.size):
            t0 = time.perf_counter()
            pool.process(x)
            t[i] = time.perf_counter() - t0
        kernel = pool.kernel_name
        pool.close()
        spec = O.load_model(path)
        secs, _ = O.cpu_bench(spec, O.default_controls(), x, n_blocks=1500, warm_blocks=20, n_threads=1, fast=True)
        out.append({"model": label, "kernel": kernel, "frames": N_FRAMES,
                    "gpu_call_us": {"p50": float(np.percentile(t, 50) * 1e6), "p99": float(np.percentile(t, 99) * 1e6),
                                    "p99.9": float(np.percentile(t, 99.9) * 1e6), "max": float(t.max() * 1e6),
                                    "calls": int(t.size), "calls_over_1.5x_p50": int((t > 1.5 * np.percentile(t, 50)).sum()),
                                    **slow_call_pattern(t)},
                    "gpu_realtime_factor": float((N_FRAMES / 48000.0) / np.percentile(t, 50)),
                    "cpu_one_thread_block_us": secs / 1500 * 1e6,
                    "cpu_realtime_factor": float((N_FRAMES / 48000.0) / (secs / 1500))})
    return out


def _pct(t_us):
    import numpy as np
    return {"p50": float(np.percentile(t_us, 50)), "p99": float(np.percentile(t_us, 99)), "p99.9": float(np.percentile(t_us, 99.9)),
            "max": float(t_us.max()), "calls": int(t_us.size)}


def _paced(call, period_s, n_calls):
    """`call()` once per period: spin (not sleep) until the deadline, like a host's audio thread that wakes on its clock; a call that
    overruns its period starts the next one late, nothing is skipped. Returns the calls' wall times in us."""
    import numpy as np
    t = np.empty(n_calls)
    deadline = time.perf_counter() + period_s
    for i in range(n_calls):
        while time.perf_counter() < deadline:
            pass
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        t[i] = t1 - t0
        deadline = max(deadline + period_s, t1)
    return t * 1e6


def _sclk_while(fn):
    """shader clock (MHz) read by rocm-smi from a thread while fn() runs; None without rocm-smi"""
    import re
    import shutil
    import threading
    seen = {}
    if not shutil.which("rocm-smi"):
        fn()
        return None

    def sample():
        time.sleep(0.15)
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20)
            rows = [ln.split(",") for ln in r.stdout.strip().splitlines() if ln.strip()]
            if len(rows) >= 2:
                for k, v in zip(rows[0], rows[1]):
                    m = re.search(r"[-0-9.]+", v)
                    if k.startswith("sclk clock speed") and m:
                        seen["sclk_mhz"] = float(m.group(0))
                    elif "Power" in k and m:
                        seen["socket_power_w"] = float(m.group(0))
        except Exception:
            pass
    th = threading.Thread(target=sample)
    th.start()
    fn()
    th.join()
    return seen or None


def realtime_paced(ax, W, torch, local, launch_stream):
    """The reference is lv2:hardRTCapable (rt-neural-generic.ttl:25,56) and run() (rt-neural-generic.cpp:484) is called once per
    audio period, not back to back: between two calls the GPU idles, its clocks fall, the host thread's caches cool. Every case here is
    measured both ways in the same process: back to back (what realtime_case reports) and PACED at the period the block length implies."""
    import numpy as np
    out = {"how": "one call per period, the caller spins until the deadline (time.perf_counter); back_to_back = the same call in a tight loop; "
                  "kernel_us = the pass's own duration from HIP events on the launch stream", "cases": []}
    bundled = os.path.join(ROOT, "tests", "golden", "models", "tw40_california_clean_deerinkstudios.json")
    path, _j = workload_model_path(W, "cfg2")
    S = WORKLOADS["cfg2"]["streams"]

    def lv2_cases(frames_list, cases):
        # (1) the pool an LV2 instance owns: one stream, the bundled LSTM-12
        pool = ax.Pool(1, 8192, 48000.0, device=local)
        pool.set_model(ax.Model(bundled))
        for frames in frames_list:
            x = W.signal(1, frames, seed=5)
            period = frames / 48000.0
            for _ in range(200):
                pool.process(x)
            b2b = np.array([0.0] * 1500)
            for i in range(b2b.size):
                t0 = time.perf_counter()
                pool.process(x)
                b2b[i] = (time.perf_counter() - t0) * 1e6
            n_calls = int(min(1500, max(300, 2.0 / period)))
            holder = {}
            state = _sclk_while(lambda: holder.__setitem__("t", _paced(lambda: pool.process(x), period, n_calls)))
            cases.append({"case": "one LV2 instance (one-stream pool, bundled LSTM-12), aidax_pool_process", "frames": frames,
                          "period_us": period * 1e6, "kernel": pool.kernel_name, "paced_us": _pct(holder["t"]), "back_to_back_us": _pct(b2b),
                          "gpu_while_paced": state})
        pool.close()

    def cfg2_cases(frames_list, cases):
        # (2) cfg2's pool, device-resident blocks, one launch per period: the kernel's duration when it arrives after the idle gap
        pool = ax.Pool(S, N_FRAMES, 48000.0, device=local)
        pool.set_model(ax.Model(path))
        pool.set_controls(ax.default_controls())
        for frames in frames_list:
            d_in = torch.from_numpy(W.signal(S, frames, seed=77)).cuda()
            d_out = torch.empty_like(d_in)
            period = frames / 48000.0
            n_calls = int(min(1000, max(300, 2.0 / period)))
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_calls)]
            idx = [0]

            def call():
                e0, e1 = evs[idx[0] % n_calls]
                e0.record(launch_stream)
                pool.process_device(d_in.data_ptr(), d_out.data_ptr(), frames, launch_stream.cuda_stream)
                e1.record(launch_stream)
                launch_stream.synchronize()
                idx[0] += 1
            for _ in range(50):
                call()
            idx[0] = 0
            b2b_wall = np.empty(n_calls)
            for i in range(n_calls):
                t0 = time.perf_counter()
                call()
                b2b_wall[i] = (time.perf_counter() - t0) * 1e6
            b2b_kern = np.array([a_.elapsed_time(b_) * 1e3 for a_, b_ in evs])
            idx[0] = 0
            holder = {}
            state = _sclk_while(lambda: holder.__setitem__("t", _paced(call, period, n_calls)))
            paced_kern = np.array([a_.elapsed_time(b_) * 1e3 for a_, b_ in evs])
            cases.append({"case": f"cfg2 pool ({S} streams, LSTM-32), aidax_pool_process_device + stream sync, blocks resident in HBM", "frames": frames,
                          "period_us": period * 1e6, "kernel": pool.kernel_name, "paced_us": _pct(holder["t"]), "back_to_back_us": _pct(b2b_wall),
                          "kernel_us_paced": _pct(paced_kern), "kernel_us_back_to_back": _pct(b2b_kern), "gpu_while_paced": state})
        pool.close()

    lv2_cases((64, 128, 256), out["cases"])
    cfg2_cases((64, 128, 256), out["cases"])
    # ... and the same paced calls with the library's keep-warm thread on (AIDAX_KEEP_WARM_US: an empty grid on a lowest-priority stream
    # every 500 us while a pool exists): what it buys a host that calls once per period
    os.environ["AIDAX_KEEP_WARM_US"] = "500"
    try:
        kw = []
        lv2_cases((64, 256), kw)
        cfg2_cases((64, 256), kw)
        out["keep_warm"] = {"AIDAX_KEEP_WARM_US": 500, "cases": kw}
    finally:
        del os.environ["AIDAX_KEEP_WARM_US"]
    # (3) a hub of 1024 seats (many plugin instances of one process, one pass per period): one aidax_hub_run per seat and
    # period from this (python) thread — the round of 1024 calls is what the host pays per period; the pass runs under the next round
    seats = 1024
    hub = ax.Hub(seats, N_FRAMES, 48000.0, device=local)
    hub.set_model(ax.Model(path))
    slots = [hub.attach() for _ in range(seats)]
    xs = W.signal(seats, N_FRAMES, seed=99)
    rows = [np.ascontiguousarray(xs[i]) for i in range(seats)]

    def round_of_runs():
        for i in range(seats):
            hub.run(slots[i], rows[i])
    for _ in range(5):
        round_of_runs()
    period = N_FRAMES / 48000.0
    b2b = np.empty(60)
    for i in range(b2b.size):
        t0 = time.perf_counter()
        round_of_runs()
        b2b[i] = (time.perf_counter() - t0) * 1e6
    holder = {}
    state = _sclk_while(lambda: holder.__setitem__("t", _paced(round_of_runs, period, 200)))
    hub.flush()
    out["cases"].append({"case": f"hub of {seats} seats (cfg2's model), {seats} aidax_hub_run calls per period from one python thread", "frames": N_FRAMES,
                         "period_us": period * 1e6, "paced_us": _pct(holder["t"]), "back_to_back_us": _pct(b2b),
                         "launches": int(hub.launches), "deadline_launches": int(hub.deadline_launches), "gpu_while_paced": state,
                         "note": "a round = every seat's run(): submit this period's block, collect the last one's (one period of latency); "
                                 "the figure is the host's time per period, ctypes overhead of 1024 python calls included"})
    hub.close()
    return out


def host_inclusive(ax, W, local):
    """cfg2 blocks from and to HOST memory, the three ways include/aidax.h offers (never `value`: the headline's inputs are resident in
    HBM). process: pageable buffers, blocking. submit/collect: one block in flight — upload of k+1 and download of k-1 under the pass of
    k. submit_to: the caller's buffers registered (page-locked once), two in/out pairs alternating, no staging copy."""
    import numpy as np
    path, _j = workload_model_path(W, "cfg2")
    S = WORKLOADS["cfg2"]["streams"]
    n_blocks = 400
    res = {"workload": WORKLOADS["cfg2"]["text"], "blocks": n_blocks, "forms": []}

    def entry(name, secs):
        us = secs / n_blocks * 1e6
        res["forms"].append({"form": name, "us_per_block": us, "samples_per_s": S * N_FRAMES / (us * 1e-6)})
    import ctypes as C
    L = ax.lib()
    fp = C.POINTER(C.c_float)

    def ptr(a):
        return a.ctypes.data_as(fp)

    def ok(rc):
        if rc < 0:
            raise SystemExit(f"bench.py: host_inclusive: {L.aidax_last_error().decode(errors='replace')}")
    # one page-locked-able arena: four input blocks, four output blocks; the pointers are made once (the loops below are the C ABI's calls and
    # nothing else: a ctypes call with ready arguments costs ~1 us, what a C host pays plus that)
    arena = np.zeros((8, S, N_FRAMES), np.float32)
    for r in range(4):
        arena[r] = W.signal(S, N_FRAMES, seed=0xA1DA + r)
    pin, pout = [ptr(arena[r]) for r in range(4)], [ptr(arena[4 + r]) for r in range(4)]
    pool = ax.Pool(S, N_FRAMES, 48000.0, device=local)
    pool.set_model(ax.Model(path))
    pool.set_controls(ax.default_controls())
    h = pool.h
    for k in range(20):
        ok(L.aidax_pool_process(h, pin[k % 4], pout[k % 4], N_FRAMES))
    t0 = time.perf_counter()
    for k in range(n_blocks):
        ok(L.aidax_pool_process(h, pin[k % 4], pout[k % 4], N_FRAMES))
    entry("aidax_pool_process (pageable host buffers, blocking)", time.perf_counter() - t0)
    ok(L.aidax_pool_submit(h, pin[0], N_FRAMES))
    for k in range(1, 20):
        ok(L.aidax_pool_submit(h, pin[k % 4], N_FRAMES))
        ok(L.aidax_pool_collect(h, pout[(k - 1) % 4], N_FRAMES))
    t0 = time.perf_counter()
    for k in range(20, 20 + n_blocks):
        ok(L.aidax_pool_submit(h, pin[k % 4], N_FRAMES))
        ok(L.aidax_pool_collect(h, pout[(k - 1) % 4], N_FRAMES))
    entry("aidax_pool_submit + collect (pageable, one block in flight)", time.perf_counter() - t0)
    ok(L.aidax_pool_collect(h, pout[(20 + n_blocks - 1) % 4], N_FRAMES))
    # registered: the arena page-locked once; in / out pairs alternate (a block's buffers are the pool's from submit_to to its collect)
    pool.register_host(arena)
    ok(L.aidax_pool_submit_to(h, pin[0], pout[0], N_FRAMES))
    for k in range(1, 20):
        ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N_FRAMES))
        ok(L.aidax_pool_collect(h, pout[(k - 1) % 4], N_FRAMES))
    t0 = time.perf_counter()
    for k in range(20, 20 + n_blocks):
        ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N_FRAMES))
        ok(L.aidax_pool_collect(h, pout[(k - 1) % 4], N_FRAMES))
    entry("aidax_pool_submit_to + collect (registered host buffers, one block in flight)", time.perf_counter() - t0)
    ok(L.aidax_pool_collect(h, pout[(20 + n_blocks - 1) % 4], N_FRAMES))
    # ... and with TWO blocks kept in flight (the pool has three staging sets): submit(k) before collect(k - 2) — the host's own round trip
    # (a download, its turn-around, an upload) is then off the GPU's critical path
    ok(L.aidax_pool_submit_to(h, pin[0], pout[0], N_FRAMES))
    ok(L.aidax_pool_submit_to(h, pin[1], pout[1], N_FRAMES))
    for k in range(2, 20):
        ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N_FRAMES))
        ok(L.aidax_pool_collect(h, pout[(k - 2) % 4], N_FRAMES))
    t0 = time.perf_counter()
    for k in range(20, 20 + n_blocks):
        ok(L.aidax_pool_submit_to(h, pin[k % 4], pout[k % 4], N_FRAMES))
        ok(L.aidax_pool_collect(h, pout[(k - 2) % 4], N_FRAMES))
    entry("aidax_pool_submit_to + collect (registered host buffers, two blocks in flight)", time.perf_counter() - t0)
    for k in (20 + n_blocks - 2, 20 + n_blocks - 1):
        ok(L.aidax_pool_collect(h, pout[k % 4], N_FRAMES))
    pool.unregister_host(arena)
    pool.close()
    return res


def slow_call_pattern(t):
    """What THIS run's slow calls (over 1.5 x p50) look like: how far apart (in calls and in wall time) and how long — measured, not
    quoted. A fixed spacing in wall time points at a timer on the host side of the spin-wait (profiles/r04_rt_latency.txt found one
    every ~10 ms on that box), a random one at the scheduler."""
    import numpy as np
    p50 = float(np.percentile(t, 50))
    idx = np.nonzero(t > 1.5 * p50)[0]
    out = {"slow_calls_us_p50": float(np.median(t[idx]) * 1e6) if idx.size else None}
    if idx.size >= 3:
        gaps = np.diff(idx)
        start = np.concatenate([[0.0], np.cumsum(t)[:-1]])
        wall = np.diff(start[idx])
        out.update({"slow_call_gap_calls_p50": float(np.median(gaps)), "slow_call_gap_ms_p50": float(np.median(wall) * 1e3),
                    "slow_call_gap_ms_iqr": [float(np.percentile(wall, 25) * 1e3), float(np.percentile(wall, 75) * 1e3)]})
    return out


def measure(ax, W, torch, name, S, steps, warmup, rank, world, local, check, launch_stream, preroll_s=PREROLL_S, dist_steps=True):
    """One timed region of one workload on this rank. Returns a dict with the local elapsed time, the average pass
    duration from HIP events on the launch stream, the pool's kernel name, the parity spot-check and the json."""
    import numpy as np
    wl = WORKLOADS[name]
    path, j = workload_model_path(W, name)
    model = ax.Model(path)
    pool = ax.Pool(S, N_FRAMES, 48000.0, device=local)
    pool.set_model(model, ax.START_WARMUP)
    pool.set_controls(ax.default_controls(**wl["controls"]))

    # synthetic inputs of this rank's stream range, resident in HBM before timing starts
    lo, _hi = stream_range(rank, world, S * world)
    host_ring = [W.signal(S, N_FRAMES, seed=0xA1DA + 7919 * r + lo) for r in range(RING)]
    d_in = [torch.from_numpy(b).cuda() for b in host_ring]
    d_out = [torch.empty_like(t) for t in d_in]
    stream = launch_stream.cuda_stream

    def step(i):
        k = i % RING
        pool.process_device(d_in[k].data_ptr(), d_out[k].data_ptr(), N_FRAMES, stream)

    # parity spot-check (outside the timed region) on the pool that is timed, through the kernel that is timed: its first blocks,
    # sixteen streams spread over the pool, against the CPU oracle on the same inputs
    max_err, parity = None, None
    if rank == 0 and check:
        from oracle import oracle as O
        nb = 4 if name != "cfg5" else 2
        rows = sorted(set(int(round(k * (S - 1) / 15.0)) for k in range(16)))
        got = []
        for k in range(nb):
            step(k)
            launch_stream.synchronize()
            got.append(d_out[k][rows].cpu().numpy())
        got = np.concatenate(got, axis=1)
        xs = np.concatenate([b[rows] for b in host_ring[:nb]], axis=1)
        want = O.run_streams(O.parse_model(j), O.default_controls(**wl["controls"]), xs, N_FRAMES)
        max_err = float(np.abs(got - want).max())
        parity = {"max_abs_err": max_err, "kernel": pool.kernel_name, "pool": "the timed pool (its first blocks, before the pre-roll)",
                  "streams_checked": rows, "blocks": nb, "against": "CPU oracle (oracle/), same inputs", "bar": 1e-5}

    # clock pre-roll: un-timed, not part of `warmup`; a short driver run then sees the same clocks as a long one
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < preroll_s:
        for _ in range(32):
            step(i)
            i += 1
        launch_stream.synchronize()
    preroll_ms = (time.perf_counter() - t0) * 1e3

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    import torch.distributed as dist
    in_group = dist.is_available() and dist.is_initialized()       # N > 1, or --force-dist at N = 1
    if in_group:
        dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(launch_stream)
    for i in range(steps):
        step(i)
    ev1.record(launch_stream)
    torch.cuda.synchronize()
    if in_group:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(steps, 1)            # avg pass duration on the launch stream
    kernel = pool.kernel_name
    if parity is not None and parity["kernel"] != kernel:
        raise SystemExit(f"bench.py: the parity check ran on {parity['kernel']}, the timed region on {kernel}")
    per_launch = launch_distribution(torch, step, launch_stream, DIST_STEPS.get(name, 200)) if dist_steps else None
    gpu_state = gpu_state_while(step, launch_stream) if (dist_steps and rank == 0) else None
    pool.close()
    return dict(elapsed=elapsed, kernel_ms=kernel_ms, kernel=kernel, max_err=max_err, parity=parity, json=j, preroll_ms=preroll_ms,
                per_launch_us=per_launch, gpu_state=gpu_state)


def launch_distribution(torch, step, launch_stream, n):
    """One HIP event pair PER LAUNCH (n + 1 events on the launch stream, back to back: a launch's figure is the time from the end of
    the launch before it to its own end), after the timed region: min / p50 / p95 / max and the launches over 1.2 x p50. A kernel
    with a second operating point shows here; an average hides it."""
    import numpy as np
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(launch_stream)
    for i in range(n):
        step(i)
        ev[i + 1].record(launch_stream)
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
    p50 = float(np.percentile(t, 50))
    return {"n": n, "min": float(t.min()), "p50": p50, "p95": float(np.percentile(t, 95)), "max": float(t.max()),
            "over_1.2x_p50": int((t > 1.2 * p50).sum()), "mean": float(t.mean())}


def gpu_state_while(step, launch_stream, busy_s=0.6):
    """Shader clock and socket power WHILE this workload's kernel runs (rocm-smi from a thread, the launches from this one): cfg5
    draws the board's whole power budget, and a box that caps it earlier runs that kernel — and only that one — slower
    (profiles/r05_cfg5_modes.txt). None when rocm-smi is not there."""
    import re
    import shutil
    import threading
    if not shutil.which("rocm-smi"):
        return None
    seen = {}

    def sample():
        time.sleep(0.1)
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=20)
            rows = [ln.split(",") for ln in r.stdout.strip().splitlines() if ln.strip()]
            if len(rows) >= 2:
                for k, v in zip(rows[0], rows[1]):
                    m = re.search(r"[-0-9.]+", v)
                    if k.startswith("sclk clock speed") and m:
                        seen["sclk_mhz"] = float(m.group(0))
                    elif k.startswith("mclk clock speed") and m:
                        seen["mclk_mhz"] = float(m.group(0))
                    elif k.startswith("fclk clock speed") and m:
                        seen["fclk_mhz"] = float(m.group(0))
                    elif "Power" in k and m:
                        seen["socket_power_w"] = float(m.group(0))
                    elif "junction" in k and m:
                        seen["junction_c"] = float(m.group(0))
        except Exception:
            pass
    th = threading.Thread(target=sample)
    th.start()
    t0 = time.perf_counter()
    i = 0
    while th.is_alive() or time.perf_counter() - t0 < 0.2:
        for _ in range(16):
            step(i)
            i += 1
        launch_stream.synchronize()
        if time.perf_counter() - t0 > 25.0:
            break
    th.join()
    return seen or None


def rooflines(name, S, kernel_ms, kernel):
    wl = WORKLOADS[name]
    algo_bytes = ALGO_BYTES_PER_SAMPLE * S * N_FRAMES                 # per pass
    algo_flops = wl["flops"] * S * N_FRAMES
    gbps = algo_bytes / (kernel_ms * 1e-3) / 1e9
    tflops = algo_flops / (kernel_ms * 1e-3) / 1e12
    multi = "+" in kernel
    note = "kernel_ms spans the three launches of the split form (k_chain, model kernel, k_chain)" if multi else None
    hbm = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
           "traffic": None, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes}
    comp = {"bound": "mfma" if wl["bound"] == "mfma" else "fp32", "achieved": tflops, "peak": FP32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tflops / FP32_PEAK_TFLOPS, "traffic": None, "kernel_ms": kernel_ms,
            "algorithmic_flops_per_launch": algo_flops}
    if ("k_gru_gs" in kernel or "k_mfma_ls" in kernel or "k_conv_ms" in kernel or "k_conv_st" in kernel) and wl.get("split_flops"):
        # The contraction runs on the bf16 matrix pipe: every fp32 product as SPLIT_PRODUCTS bf16 term products of operands
        # split exactly into three bf16 terms (fp32 MFMAs run at the vector rate on gfx950 and stall the VALU beside them,
        # profiles/r04_overlap.txt). What binds the kernel is then the bf16 MFMA peak against the matrix flops it EXECUTES
        # (SPLIT_PRODUCTS x the algorithmic flops of the split products + the rest at face value); the algorithmic fp32 rate stays in the entry.
        executed = (SPLIT_PRODUCTS * wl["split_flops"] + (wl["flops"] - wl["split_flops"])) * S * N_FRAMES
        ex_tflops = executed / (kernel_ms * 1e-3) / 1e12
        comp = {"bound": "mfma", "achieved": ex_tflops, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ex_tflops / BF16_PEAK_TFLOPS,
                "traffic": None, "kernel_ms": kernel_ms, "executed_flops_per_launch": executed, "algorithmic_flops_per_launch": algo_flops,
                "algorithmic_tflops": tflops, "algorithmic_frac_of_fp32_mfma_peak": tflops / FP32_PEAK_TFLOPS,
                "arithmetic": f"fp32 operands split exactly into three bf16 terms, {SPLIT_PRODUCTS} bf16 term products per fp32 product "
                              f"(v_mfma_f32_16x16x32_bf16, fp32 accumulation); achieved = executed matrix flops against the dense bf16 peak"}
    if note:
        hbm["note"] = comp["note"] = note
    return hbm, comp


def dry_run(args):
    """CPU stand-in for the multi-rank plumbing (tests): same rank logic and the same reduction over gloo, the
    'pass' is a sleep. Never used for a reported number."""
    import torch
    import torch.distributed as dist
    rank, world, _local = dist_env()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    S = args.streams or WORKLOADS[args.workload]["streams"]
    for _ in range(args.warmup):
        time.sleep(0.0005)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max, samples_all = reduce_results(elapsed, float(S) * N_FRAMES * args.steps, world, backend_device="cpu")
    lo, hi = stream_range(rank, world, S * world)
    if world > 1:
        ranges = [None] * world
        dist.all_gather_object(ranges, (lo, hi))
        dist.destroy_process_group()
    else:
        ranges = [(lo, hi)]
    if rank == 0:
        print(json.dumps({"metric": "audio samples/sec (48 kHz mono, many streams)", "value": samples_all / elapsed_max,
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": WORKLOADS[args.workload]["text"], "streams_per_gpu": S, "frames": N_FRAMES,
                                     "stream_ranges": ranges}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the workload's, cfg2 = 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short cfg3/cfg4/cfg5 regions and the one-stream case")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic (the two rocprofv3 --pmc child runs)")
    ap.add_argument("--no-dist", action="store_true", help="skip the per-launch distribution and the clock / power sample after each timed region")
    ap.add_argument("--dry-run", action="store_true", help="CPU stand-in for the rank plumbing (gloo, no GPU): tests only")
    ap.add_argument("--share-device", action="store_true",
                    help="rank plumbing with the real kernels on a box with fewer GPUs than ranks: every rank runs on device 0, "
                         "the reduction goes over gloo; the line says so and is no scaling measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path whatever N is: ranks started under torch.distributed.run from the GPU-less parent, "
                         "init_process_group('nccl'), the reductions as RCCL collectives on cuda tensors, cfg4 / cfg5 regions per rank "
                         "(--gpus 1 --force-dist is how the multi-GPU branch runs on a one-GPU box)")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.force_dist):
        # before anything touches the GPU: this process only starts the ranks and relays their line
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)

    import torch
    rank, world, local = dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    ax = importlib.import_module("aidadsp-lv2_amd")
    W = ax.workloads
    wl = WORKLOADS[args.workload]
    S = args.streams or wl["streams"]
    launch_stream = torch.cuda.Stream()              # a real (non-null) HIP stream: the events sit on it too
    torch.cuda.set_stream(launch_stream)

    m = measure(ax, W, torch, args.workload, S, args.steps, args.warmup, rank, world, local, not args.no_check, launch_stream,
                dist_steps=not args.no_dist)
    samples = float(S) * N_FRAMES * args.steps
    red_dev = "cpu" if args.share_device else "cuda"
    elapsed_max, samples_all = reduce_results(m["elapsed"], samples, world, backend_device=red_dev, force=use_dist)
    rank_elapsed = [m["elapsed"]]
    if use_dist:
        mine = torch.tensor([m["elapsed"]], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_elapsed = [float(t.item()) for t in every]

    # N > 1: BASELINE's 8-GPU workloads (cfg4: 1024 streams per GPU, cfg5: 2048) on EVERY rank, each region bracketed
    # and reduced like the headline (not with --share-device: ranks that share one GPU are rank plumbing, and two
    # processes cannot both hold the CUs k_mfma_lp needs)
    multi_others = []
    if use_dist and not args.no_others and not args.share_device:
        for name in ("cfg4", "cfg5"):
            if name == args.workload:
                continue
            So, steps = WORKLOADS[name]["streams"], OTHER_STEPS[name]
            r = measure(ax, W, torch, name, So, steps, max(10, steps // 10), rank, world, local, not args.no_check, launch_stream,
                        preroll_s=0.15, dist_steps=not args.no_dist)
            e_max, n_all = reduce_results(r["elapsed"], float(So) * N_FRAMES * steps, world, backend_device=red_dev, force=use_dist)
            multi_others.append((name, So, steps, r, e_max, n_all))

    if rank == 0:
        value = samples_all / elapsed_max
        hbm, comp = rooflines(args.workload, S, m["kernel_ms"], m["kernel"])
        traffic, how = None, None
        if args.workload == "cfg2" and S == wl["streams"] and not args.no_traffic and "+" not in m["kernel"]:
            if not use_dist:
                traffic, how = measure_traffic_live(args.workload, m["kernel"], 4.0 * S * N_FRAMES), "measured by this run"
            if traffic is None:
                traffic, how = pmc_traffic_bytes(m["kernel"], 4.0 * S * N_FRAMES), "committed profile of the same kernel sources, not measured in this run"
        hbm["kernel_src_sha16"], hbm["lib_sha16"] = kernel_sources_sha16(), lib_sha16(ax)
        if traffic:
            hbm["traffic"] = traffic["bytes"]
            hbm["traffic_source"] = (f"{how}: {traffic['source']}; FETCH_SIZE {traffic['fetch_size_kb']:.0f} KB + half of the 16 B/lane "
                                     f"audio read it under-reports + WRITE_SIZE {traffic['write_size_kb']:.0f} KB; algorithmic 8 B/sample + "
                                     f"per-stream control/state/NN records = 3.28 MB")
        out = {
            "metric": "audio samples/sec (48 kHz mono, many streams)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["text"],
                       "streams_per_gpu": S, "frames": N_FRAMES, "kernel": m["kernel"],
                       "realtime_factor": value / (48000.0 * S * world)},
            "roofline": comp if wl["bound"] == "mfma" else hbm,
            "roofline_compute": comp,
            "preroll_ms": m["preroll_ms"],
            "max_abs_err": m["max_err"], "max_abs_err_kernel": m["parity"]["kernel"] if m["parity"] else None, "parity": m["parity"],
            "per_launch_us": m["per_launch_us"], "gpu_state_while_running": m["gpu_state"],
            "ranks": {"world_size": world, "backend": ("gloo" if args.share_device else "nccl (RCCL)") if use_dist else None,
                      "elapsed_s": rank_elapsed, "collective": "all_reduce MAX(elapsed) + SUM(samples), all_gather(elapsed): after the timed region only"},
        }
        if wl["bound"] == "mfma":
            out["roofline_hbm"] = hbm
        if args.share_device:
            out["share_device"] = True                        # all ranks on device 0: the rank plumbing with real kernels, not a scaling number
            out["config"]["note"] = f"{world} ranks share ONE GPU (--share-device): no scaling measurement"
        if not use_dist and not args.no_others:
            others = []
            for name in ("cfg3", "cfg4", "cfg5"):
                if name == args.workload:
                    continue
                So, steps = WORKLOADS[name]["streams"], OTHER_STEPS[name]
                r = measure(ax, W, torch, name, So, steps, max(10, steps // 10), 0, 1, local, not args.no_check, launch_stream,
                            preroll_s=0.15, dist_steps=not args.no_dist)
                h2, c2 = rooflines(name, So, r["kernel_ms"], r["kernel"])
                add_profile_figures(h2, c2, name, So, r["kernel"], r["kernel_ms"])
                others.append({"workload": WORKLOADS[name]["text"], "streams": So, "kernel": r["kernel"], "steps": steps,
                               "ms_per_step": r["elapsed"] / steps * 1e3, "value": So * N_FRAMES * steps / r["elapsed"],
                               "unit": "samples/s", "roofline": c2 if WORKLOADS[name]["bound"] == "mfma" else h2,
                               "roofline_compute": c2, "max_abs_err": r["max_err"],
                               "max_abs_err_kernel": r["parity"]["kernel"] if r["parity"] else None, "parity": r["parity"],
                               "per_launch_us": r["per_launch_us"], "gpu_state_while_running": r["gpu_state"]})
                if WORKLOADS[name]["bound"] == "mfma":
                    others[-1]["roofline_hbm"] = h2
                if not args.no_cpu_baseline:
                    others[-1]["cpu_baseline"] = cpu_baseline(W, r["json"], name, target_s=3.0)
            out["other_workloads"] = others
            if args.workload == "cfg2":
                # the same model with more streams per GPU than BASELINE's 1024: where the recurrent waves stop being alone
                # on their SIMDs and throughput, not latency, is what is measured (the pool picks the form per size)
                sweep = []
                for So, steps in ((4096, 400), (16384, 120)):
                    r = measure(ax, W, torch, "cfg2", So, steps, max(10, steps // 10), 0, 1, local, False, launch_stream, preroll_s=0.1, dist_steps=False)
                    sweep.append({"streams": So, "kernel": r["kernel"], "steps": steps, "ms_per_step": r["elapsed"] / steps * 1e3,
                                  "value": So * N_FRAMES * steps / r["elapsed"], "unit": "samples/s"})
                out["stream_sweep"] = sweep
            out["realtime_case"] = realtime_case(ax, W, local)
            out["realtime_paced"] = realtime_paced(ax, W, torch, local, launch_stream)
            out["host_inclusive"] = host_inclusive(ax, W, local)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, m["json"], args.workload)
        if multi_others:
            out["other_workloads"] = []
            for name, So, steps, r, e_max, n_all in multi_others:
                h2, c2 = rooflines(name, So, r["kernel_ms"], r["kernel"])
                add_profile_figures(h2, c2, name, So, r["kernel"], r["kernel_ms"])
                out["other_workloads"].append({"workload": WORKLOADS[name]["text"], "streams_per_gpu": So, "n_gpus": world, "kernel": r["kernel"],
                                               "steps": steps, "ms_per_step": e_max / steps * 1e3, "value": n_all / e_max, "unit": "samples/s",
                                               "scaling": "weak", "roofline": c2 if WORKLOADS[name]["bound"] == "mfma" else h2,
                                               "roofline_compute": c2, "max_abs_err": r["max_err"],
                                               "max_abs_err_kernel": r["parity"]["kernel"] if r["parity"] else None, "parity": r["parity"],
                                               "per_launch_us": r["per_launch_us"], "gpu_state_while_running": r["gpu_state"]})

    if use_dist:
        dist.destroy_process_group()       # RCCL prints its banner on teardown: keep the JSON line last
    if rank == 0:
        # RCCL writes its version banner to stdout through C stdio (it would otherwise surface at exit, after
        # our line): drain both buffers first so that the JSON line is the last thing on stdout
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
