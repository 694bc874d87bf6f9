"""world_size-2 gloo test (CPU) of the only multi-rank logic on the path: contiguous
stream sharding and the final MAX(elapsed)/SUM(samples) reduction of bench.py. The
data path itself has no collective (streams are independent)."""
import os
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
