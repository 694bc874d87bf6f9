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

N_STREAMS = 1024          # per GPU (cfg2)
N_FRAMES = 256
RING = 8                  # distinct input blocks cycled through HBM
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector FP32 peak
ALGO_BYTES_PER_SAMPLE = 8


FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD

# The headline line is cfg2 (BASELINE.json configs[1]). The other GPU configs can be timed with
# --workload; they are parity-test cases, not the bench line the driver records.
WORKLOADS = {
    "cfg2": dict(model=dict(kind="lstm", hidden=32, input_size=1, seed=32), streams=1024, controls={},
                 flops=8512 + 63 + 6, bound="hbm",
                 text="cfg2: LSTM-32 amp model, 1024 streams/GPU x 256-frame blocks, full run() chain, TTL-default controls"),
    "cfg3": dict(model=dict(kind="gru", hidden=64, input_size=3, seed=64), streams=4096,
                 controls=dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0, depth_boost_db=3.0,
                               presence_boost_db=3.0, param1=0.5, param2=0.3),
                 flops=2 * (3 * 64 * (64 + 3) + 64) + 63 + 6, bound="hbm",
                 text="cfg3: GRU-64 conditioned (PARAM1+PARAM2) pedal model + 5-band EQ post, 4096 streams x 256-frame blocks"),
    "cfg4": dict(model=dict(kind="conv", hidden=16, input_size=1, seed=1608), streams=1024, controls={},
                 flops=2 * (3 * 1 * 16 + 7 * 3 * 16 * 16 + 16) + 63 + 6, bound="hbm",
                 text="cfg4: dilated conv1d stack (8 layers, 16 channels, k=3), 1024 streams/GPU (8192 over 8 GPUs) x 256-frame blocks"),
    "cfg5": dict(model=dict(kind="lstm", hidden=96, input_size=1, seed=96, n_rnn=2), streams=2048, controls={},
                 flops=2 * (384 * (1 + 96) + 384 * (96 + 96) + 96), bound="mfma",
                 text="cfg5: LSTM-96 x2 model, 2048 streams/GPU (16384 over 8 GPUs) x 256-frame blocks, matrix-core kernel"),
}


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


def workload_model_path(name: str):
    from tests import modelgen
    j = modelgen.make_model(**WORKLOADS[name]["model"])
    d = tempfile.mkdtemp(prefix="aidax_bench_")
    return modelgen.write_model(j, os.path.join(d, f"{name}.json")), j


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


def cpu_baseline(j, name: str = "cfg2", target_s: float = 12.0):
    """The CPU oracle on all host cores over a bounded sample of the workload."""
    from oracle import oracle as O
    from tests import modelgen
    spec = O.parse_model(j)
    cores = usable_cores()
    streams = N_STREAMS                      # 1024 of the workload's streams; the sample is bounded in blocks
    x = modelgen.signal(streams, N_FRAMES)
    c = O.default_controls(**WORKLOADS[name]["controls"])
    secs, _ = O.cpu_bench(spec, c, x, n_blocks=2, warm_blocks=1, n_threads=cores, fast=True)
    per_block = secs / 2
    blocks = int(max(4, min(20000, target_s / max(per_block, 1e-6))))
    secs, _ = O.cpu_bench(spec, c, x, n_blocks=blocks, warm_blocks=1, n_threads=cores, fast=True)
    sps = streams * N_FRAMES * blocks / secs
    # the LV2 real-time case of SURVEY §8(d): one stream on one thread, 256-frame blocks (about a second of CPU)
    one_blocks = int(max(50, min(4000, 1.0 / max(per_block / streams * cores, 1e-7))))
    one_secs, _ = O.cpu_bench(spec, c, x[:1], n_blocks=one_blocks, warm_blocks=2, n_threads=1, fast=True)
    one_sps = N_FRAMES * one_blocks / one_secs
    return {"value": sps, "unit": "samples/s", "cores": cores, "kind": "port",
            "one_stream_one_thread": {"value": one_sps, "unit": "samples/s", "realtime_factor": one_sps / 48000.0},
            "sample": f"{streams} {name} streams x {blocks} blocks of {N_FRAMES} frames, "
                      f"full run() chain, C oracle with vectorised exp/tanh (-O3 -march=x86-64-v3, AVX2), {cores} pthreads, {secs:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the workload's, cfg2 = 1024)")
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

    wl = WORKLOADS[args.workload]
    S = args.streams or wl["streams"]
    path, j = workload_model_path(args.workload)
    model = ax.Model(path)
    pool = ax.Pool(S, N_FRAMES, 48000.0, device=local)
    pool.set_model(model, ax.START_WARMUP)
    pool.set_controls(ax.default_controls(**wl["controls"]))

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
        chk.set_controls(ax.default_controls(**wl["controls"]))
        xs = np.concatenate([b[:16] for b in host_ring[:4]], axis=1)
        got = np.concatenate([chk.process(np.ascontiguousarray(xs[:, k * N_FRAMES:(k + 1) * N_FRAMES])) for k in range(4)], axis=1)
        want = O.run_streams(O.parse_model(j), O.default_controls(**wl["controls"]), xs, N_FRAMES)
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
        traffic = pmc_traffic_bytes(pool.kernel_name) if (args.workload == "cfg2" and S == N_STREAMS) else None
        tflops = wl["flops"] * S * N_FRAMES / (kernel_ms * 1e-3) / 1e12
        if wl["bound"] == "mfma":
            roofline = {"bound": "mfma", "achieved": tflops, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tflops / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "kernel_ms": kernel_ms,
                        "algorithmic_flops_per_launch": wl["flops"] * S * N_FRAMES,
                        "note": "kernel_ms spans the three launches of the split form (k_chain, k_mfma, k_chain)"}
        else:
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS,
                        "traffic": traffic["bytes"] if traffic else None,
                        "traffic_source": traffic["source"] if traffic else None,
                        "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes}
        out = {
            "metric": "audio samples/sec (48 kHz mono, many streams)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["text"],
                       "streams_per_gpu": S, "frames": N_FRAMES, "kernel": pool.kernel_name,
                       "realtime_factor": value / (48000.0 * S * world)},
            "roofline": roofline,
            "compute": {"fp32_tflops": tflops, "peak_tflops": FP32_PEAK_TFLOPS},
            "max_abs_err": max_err,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(j, args.workload)

    pool.close()
    if world > 1:
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
