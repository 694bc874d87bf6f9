#!/usr/bin/env python3
"""bench.py — headline benchmark of the rt-neural-generic hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], "cfg2"): LSTM hidden=32 amp model (synthetic
Glorot-style weights, seed 32 — no checkpoints offline), 1024 concurrent 48 kHz
mono streams PER GPU, 256-frame blocks, plugin controls at their TTL defaults
(LPF 66.216 %, DC blocker on, EQ post flat, gains 0 dB): the full run() chain
of rt-neural-generic.cpp:621-659. One "step" = one aidax_pool_process_device
pass over one [1024][256] fp32 block already resident in HBM.

Streams are independent, so N GPUs run N x 1024 streams with no data-path
collective ("weak" scaling); RCCL carries one MAX(elapsed) / SUM(samples)
reduction at the end.

Prints ONE json line (rank 0) with the driver's contract keys plus
  roofline     HBM roofline of the dominant kernel from ALGORITHMIC bytes
               (8 B per sample per stream: 4 in + 4 out) over the
               average launch duration measured with HIP events on the launch stream
  cpu_baseline the CPU oracle (a port, not RTNeural) timed on this host's cores
               on a bounded sample of the same workload (N=1, rank 0 only)
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_STREAMS = 1024          # per GPU
N_FRAMES = 256
RING = 8                  # distinct input blocks cycled through HBM
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector FP32 peak
ALGO_BYTES_PER_SAMPLE = 8
LSTM32_FLOPS_PER_SAMPLE = 8512 + 63 + 6   # NN MACs x2 + biquad fp64 flops + ramps


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def reduce_results(elapsed_s: float, samples: float, world: int, backend_device="cuda"):
    """MAX(elapsed) and SUM(samples) over ranks — the only collective of the path."""
    if world == 1:
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


def cfg2_model_path():
    from tests import modelgen
    j = modelgen.make_model("lstm", 32, 1, seed=32)
    d = tempfile.mkdtemp(prefix="aidax_bench_")
    return modelgen.write_model(j, os.path.join(d, "lstm32_cfg2.json")), j


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


def pmc_traffic_bytes(kernel_name: str):
    """HBM bytes per launch of the benchmarked kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.txt): (2 x FETCH_SIZE + WRITE_SIZE) KB — FETCH_SIZE counts half of a
    16 B/lane streaming read on gfx950 (MI355X_MICROARCH.md, HBM section). None if no profile."""
    import glob
    import re
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.txt"))):
        fetch = write = None
        for line in open(fn):
            if kernel_name.split("<")[0] + "<" not in line or kernel_name.split("<")[1] not in line:
                continue
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ median=([0-9.e+]+)", line)
            if m and m.group(1) == "FETCH_SIZE":
                fetch = float(m.group(2))
            elif m:
                write = float(m.group(2))
        if fetch is not None and write is not None:
            best = {"bytes": (2.0 * fetch + write) * 1024.0, "source": os.path.basename(fn)}
    return best


def cpu_baseline(j, target_s: float = 12.0):
    """The CPU oracle on all host cores over a bounded sample of cfg2."""
    from oracle import oracle as O
    from tests import modelgen
    spec = O.parse_model(j)
    cores = usable_cores()
    streams = N_STREAMS                      # the whole cfg2 stream set; the sample is bounded in blocks
    x = modelgen.signal(streams, N_FRAMES)
    c = O.default_controls()
    secs, _ = O.cpu_bench(spec, c, x, n_blocks=2, warm_blocks=1, n_threads=cores, fast=True)
    per_block = secs / 2
    blocks = int(max(4, min(20000, target_s / max(per_block, 1e-6))))
    secs, _ = O.cpu_bench(spec, c, x, n_blocks=blocks, warm_blocks=1, n_threads=cores, fast=True)
    sps = streams * N_FRAMES * blocks / secs
    return {"value": sps, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"all {streams} cfg2 streams x {blocks} blocks of {N_FRAMES} frames, "
                      f"full run() chain, C oracle with vectorised exp/tanh (-O3 -march=x86-64-v3, AVX2), {cores} pthreads, {secs:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--streams", type=int, default=N_STREAMS, help="streams per GPU (cfg2: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()

    import torch
    rank, world, local = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    ax = importlib.import_module("aidadsp-lv2_amd")
    from tests import modelgen

    S = args.streams
    path, j = cfg2_model_path()
    model = ax.Model(path)
    pool = ax.Pool(S, N_FRAMES, 48000.0, device=local)
    pool.set_model(model, ax.START_WARMUP)
    pool.set_controls(ax.default_controls())

    # synthetic inputs of this rank's stream range, resident in HBM before timing starts
    lo, hi = stream_range(rank, world, S * world)
    host_ring = [modelgen.signal(S, N_FRAMES, seed=0xA1DA + 7919 * r + lo) for r in range(RING)]
    d_in = [torch.from_numpy(b).cuda() for b in host_ring]
    d_out = [torch.empty_like(t) for t in d_in]
    launch_stream = torch.cuda.Stream()              # a real (non-null) HIP stream: events below sit on it too
    torch.cuda.set_stream(launch_stream)
    stream = launch_stream.cuda_stream

    def step(i):
        k = i % RING
        pool.process_device(d_in[k].data_ptr(), d_out[k].data_ptr(), N_FRAMES, stream)

    # parity spot-check on the first blocks (outside the timed region)
    max_err = None
    if rank == 0 and not args.no_check:
        from oracle import oracle as O
        chk = ax.Pool(16, N_FRAMES, 48000.0, device=local)
        chk.set_model(model, ax.START_WARMUP)
        xs = np.concatenate([b[:16] for b in host_ring[:4]], axis=1)
        got = np.concatenate([chk.process(np.ascontiguousarray(xs[:, k * N_FRAMES:(k + 1) * N_FRAMES])) for k in range(4)], axis=1)
        want = O.run_streams(O.parse_model(j), O.default_controls(), xs, N_FRAMES)
        max_err = float(np.abs(got - want).max())
        chk.close()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)       # avg launch duration on the launch stream

    samples = float(S) * N_FRAMES * args.steps
    elapsed_max, samples_all = reduce_results(elapsed, samples, world)

    if rank == 0:
        value = samples_all / elapsed_max
        algo_bytes = ALGO_BYTES_PER_SAMPLE * S * N_FRAMES                 # per launch
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = pmc_traffic_bytes(pool.kernel_name) if S == N_STREAMS else None
        out = {
            "metric": "audio samples/sec (48 kHz mono, many streams)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: LSTM-32 amp model, 1024 streams/GPU x 256-frame blocks, full run() chain, TTL-default controls",
                       "streams_per_gpu": S, "frames": N_FRAMES, "kernel": pool.kernel_name,
                       "realtime_factor": value / (48000.0 * S * world)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes},
            "compute": {"fp32_tflops": LSTM32_FLOPS_PER_SAMPLE * S * N_FRAMES / (kernel_ms * 1e-3) / 1e12,
                        "peak_tflops": FP32_PEAK_TFLOPS},
            "max_abs_err": max_err,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(j)
        print(json.dumps(out), flush=True)

    pool.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
