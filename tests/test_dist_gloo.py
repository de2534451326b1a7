"""N>1 path on CPU: 2 processes over gloo exercise the rank helpers bench.py uses for data-parallel eval
(image sharding, barrier, max-over-ranks of the elapsed time, sum of processed images)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "faster-orefsdet_amd"))
    import torch.distributed as dist
    from detectron2.utils import comm
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert comm.get_world_size() == world and comm.get_rank() == rank
        b, e = comm.shard_range(1060)          # the reference's val set size
        comm.synchronize()
        elapsed = comm.max_over_ranks(1.0 + rank)   # slowest rank defines the job time
        total = comm.sum_over_ranks(e - b)
        gathered = comm.all_gather((rank, b, e))
        q.put((rank, b, e, elapsed, total, gathered))
    finally:
        dist.destroy_process_group()


def test_two_rank_eval_sharding_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, b0, e0, t0, n0, g0), (r1, b1, e1, t1, n1, g1) = res
    assert (b0, e0, b1, e1) == (0, 530, 530, 1060)      # contiguous, disjoint, complete
    assert t0 == t1 == 2.0 and n0 == n1 == 1060.0
    assert g0 == g1 == [(0, 0, 530), (1, 530, 1060)]


def test_shard_range_edge_cases():
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    from detectron2.utils.comm import shard_range
    for n in (0, 1, 7, 8, 9, 1060):
        for w in (1, 2, 4, 8):
            cover = []
            for r in range(w):
                b, e = shard_range(n, r, w)
                assert 0 <= b <= e <= n
                cover += list(range(b, e))
            assert cover == list(range(n)), (n, w)
