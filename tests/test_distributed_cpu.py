"""The multi-GPU exchange rehearsed on the CPU: two gloo ranks, each
with a local (min, index), must agree on the global winner with the first-index
tie-break; the block partition must cover the grid exactly once."""
import math
import os
import socket

import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cases, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from covest_amd.grid import distributed_argmin, partition_flat_range
    res = []
    for locals_ in cases:
        v, i = locals_[rank]
        res.append(distributed_argmin(v, i))
    bounds = partition_flat_range(1000, world)
    res.append((bounds[rank], bounds[rank + 1]))
    out.put((rank, res))
    dist.destroy_process_group()


def test_two_rank_argmin_exchange():
    import torch.multiprocessing as mp
    inf, nan = math.inf, math.nan
    cases = [
        [(3.0, 10), (2.0, 600)],       # rank 1 wins
        [(2.0, 400), (2.0, 600)],      # tie: lowest flat index wins
        [(2.0, 700), (2.0, 600)],      # tie, winner on the other rank
        [(inf, -1), (5.0, 999)],       # one rank has nothing below +inf
        [(inf, -1), (inf, -1)],        # nobody has: (-1)
        [(nan, 3), (7.0, 800)],        # NaN never wins
        [(-inf, 20), (1.0, 500)],      # -inf does win
    ]
    want = [(2.0, 600), (2.0, 400), (2.0, 600), (5.0, 999), (inf, -1), (7.0, 800), (-inf, 20)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cases, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert got[rank][:-1] == want, (rank, got[rank])
    assert got[0][-1] == (0, 500) and got[1][-1] == (500, 1000)
