"""world_size-2 gloo test (CPU) of the only multi-rank logic on the path: contiguous
stream sharding and the final MAX(elapsed)/SUM(samples) reduction of bench.py. The
data path itself has no collective (streams are independent)."""
import json
import os
import subprocess
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.stream_range(rank, world, 1024 * world)
    elapsed = 0.5 + 0.25 * rank                      # rank 1 is the slow one
    samples = float((hi - lo) * 256 * 10)
    t, n = bench.reduce_results(elapsed, samples, world, backend_device="cpu")
    dist.barrier()
    q.put((rank, lo, hi, t, n))
    dist.destroy_process_group()


def test_two_rank_reduction_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, n0), (r1, lo1, hi1, t1, n1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 1024, 1024, 2048)          # contiguous, disjoint, complete
    assert t0 == t1 == pytest.approx(0.75)                        # MAX over ranks
    assert n0 == n1 == 2 * 1024 * 256 * 10                        # SUM over ranks


def test_stream_ranges_partition_any_world():
    import bench
    for world in (1, 2, 3, 4, 8):
        total = 1024 * world
        edges = [bench.stream_range(r, world, total) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        for a, b in zip(edges, edges[1:]):
            assert a[1] == b[0]


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with no RANK/WORLD_SIZE in the environment: the parent starts the two ranks as
    children (before it touches HIP or torch), waits, and relays rank 0's json line as the last line of its own
    stdout. --dry-run swaps the GPU pass for a sleep and RCCL for gloo; the rank plumbing is the real one."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "5",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                 # ONE line on stdout: rank 0's
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["scaling"] == "weak" and out["steps"] == 5
    assert out["config"]["stream_ranges"] == [[0, 1024], [1024, 2048]]
    # whole-job value: both ranks' samples over the slowest rank's time (5 sleeps of 1 ms)
    assert out["value"] == pytest.approx(2 * 1024 * 256 * 5 / (out["ms_per_step"] * 5e-3))
    assert 1.0 <= out["ms_per_step"] < 20.0


def test_eight_ranks_as_the_driver_launches_them():
    """The 8-GPU line before there is an 8-GPU node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` — the driver's own command — with --dry-run (a sleep for the
    pass, gloo for RCCL). Eight contiguous stream ranges that tile cfg2's 8192 streams, ONE line (rank 0's), whole-job value =
    all ranks' samples over the slowest rank's time. And cfg4 / cfg5, BASELINE's 8-GPU workloads: 8 x 1024 = 8192 and
    8 x 2048 = 16 384 streams."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    for workload, per_gpu in (("cfg2", 1024), ("cfg4", 1024), ("cfg5", 2048)):
        run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                              "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "4",
                              "--warmup", "1", "--workload", workload], env=env, capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, lines
        out = json.loads(lines[0])
        assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["steps"] == 4 and out["dry_run"] is True
        assert out["config"]["stream_ranges"] == [[r * per_gpu, (r + 1) * per_gpu] for r in range(8)]
        assert out["value"] == pytest.approx(8 * per_gpu * 256 * 4 / (out["ms_per_step"] * 4e-3))
        if workload != "cfg2":
            break_even = out["config"]["stream_ranges"][-1][1]
            assert break_even == {"cfg4": 8192, "cfg5": 16384}[workload]          # BASELINE.json configs[3], configs[4]


def test_bench_single_rank_dry_run_needs_no_launcher_and_fails_loudly_when_a_rank_dies():
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr[-2000:]
    assert json.loads(run.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # no GPU here: the real (non dry-run) ranks exit non-zero and so does the parent, without a json line
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                             env=env, capture_output=True, text=True, timeout=300)
        assert run.returncode != 0
        assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_two_ranks_with_the_real_kernels_on_one_gpu():
    """`bench.py --gpus 2 --share-device`: the parent starts its two ranks, both run the real pool passes (on device 0:
    a box with one GPU), the only collective — MAX(elapsed), SUM(samples) — goes over gloo, rank 0 prints the line and
    marks it as no scaling measurement. The rank plumbing of SURVEY §8(e) end to end on hardware."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "40", "--warmup", "5",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["share_device"] is True and j["steps"] == 40
    assert j["value"] > 1e8 and j["max_abs_err"] < 1e-6
    assert j["config"]["streams_per_gpu"] == 1024 and "no scaling measurement" in j["config"]["note"]


def test_force_dist_takes_the_rank_path_at_world_size_one():
    """`bench.py --gpus 1 --force-dist --dry-run`: the parent starts ONE rank under torch.distributed.run and relays its
    line — the launcher half of the N > 1 path at world size 1 (the GPU half: the test below)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dry-run", "--steps", "3",
                          "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["dry_run"] is True and out["config"]["stream_ranges"] == [[0, 1024]]


@pytest.mark.gpu
def test_the_rccl_branch_runs_at_world_size_one():
    """The multi-GPU code path of bench.py executed on the one GPU of the box: the GPU-less parent starts its rank under
    torch.distributed.run, the rank joins an `nccl` (= RCCL) process group NEXT TO libaidax_hip.so in one process, runs the
    real pools (headline + the cfg4 / cfg5 regions every rank of an N > 1 run measures), and the reductions — all_reduce
    MAX(elapsed) / SUM(samples), all_gather(elapsed) — are RCCL collectives on cuda tensors. Instances share nothing
    (rt-neural-generic.h:198-239), so this is all the communication an 8-GPU run has."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "200", "--warmup", "20",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ranks"]["world_size"] == 1 and j["ranks"]["backend"] == "nccl (RCCL)"
    assert len(j["ranks"]["elapsed_s"]) == 1 and j["ranks"]["elapsed_s"][0] > 0
    assert j["value"] > 1e9 and j["max_abs_err"] < 1e-6
    names = [o["workload"][:4] for o in j["other_workloads"]]
    assert names == ["cfg4", "cfg5"] and all(o["n_gpus"] == 1 and o["value"] > 1e8 for o in j["other_workloads"])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "bench_1rank_rccl.json"), "w") as f:
            f.write(lines[0] + "\n")


def test_bench_matches_a_profile_row_to_the_pool_s_kernel_name():
    """bench.py reads its kernel's rows out of rocprofv3 tables by the name the pool reports (`k_lstm_pipe4<32>`); the instantiation in the
    table may carry further template arguments behind the quoted ones (`k_lstm_pipe4<32, false>` since the conditioned models have a kernel
    of their own) — a matcher that missed it left `roofline.traffic` null for a whole evidence run."""
    import importlib
    bench = importlib.import_module("bench")
    rows = ["void aidax::k_lstm_pipe4<32, false>(aidax::LaunchArgs)", "void aidax::k_lstm_pipe4<32>(aidax::LaunchArgs)",
            "void aidax::k_gru_gs<4, 4, 6>(aidax::LaunchArgs, aidax::MfmaDesc)", "void aidax::k_conv_st<aidax::StGeoA, 256, true>(aidax::LaunchArgs, aidax::ConvDesc)"]
    assert bench._kernel_matches("k_lstm_pipe4<32>", rows[0]) and bench._kernel_matches("k_lstm_pipe4<32>", rows[1])
    assert not bench._kernel_matches("k_lstm_pipe<32>", rows[0]) and not bench._kernel_matches("k_lstm_pipe4<3>", rows[0])
    assert not bench._kernel_matches("k_lstm_pipe4<16>", rows[0])
    assert bench._kernel_matches("k_gru_gs", rows[2]) and bench._kernel_matches("k_conv_st", rows[3]) and not bench._kernel_matches("k_conv_ms", rows[3])
