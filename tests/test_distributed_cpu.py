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
    import torch
    res = []
    for n, locals_ in enumerate(cases):
        v, i = locals_[rank]
        if n % 2:  # the pair already in a tensor, as the arg-min kernel leaves it (DenseGrid.argmin_pair_tensor)
            res.append(distributed_argmin(None, None, pair=torch.tensor([v, float(i)], dtype=torch.float64)))
        else:
            res.append(distributed_argmin(v, i))
    bounds = partition_flat_range(1000, world)
    res.append((bounds[rank], bounds[rank + 1]))
    out.put((rank, res))
    dist.destroy_process_group()


def _run(world, cases):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_rank_argmin_exchange():
    inf, nan = math.inf, math.nan
    cases = [
        [(3.0, 10), (2.0, 600)],       # rank 1 wins
        [(2.0, 400), (2.0, 600)],      # tie: lowest flat index wins
        [(2.0, 700), (2.0, 600)],      # tie, winner on the other rank
        [(inf, -1), (5.0, 999)],       # one rank has nothing below +inf
        [(inf, -1), (inf, -1)],        # nobody has: (-1)
        [(nan, 3), (7.0, 800)],        # NaN never wins
        [(-inf, 20), (1.0, 500)],      # -inf does win
        [(nan, 3), (nan, 800)],        # only NaNs: nobody
    ]
    want = [(2.0, 600), (2.0, 400), (2.0, 600), (5.0, 999), (inf, -1), (7.0, 800), (-inf, 20), (inf, -1)]
    got = _run(2, cases)
    for rank in (0, 1):
        assert got[rank][:-1] == want, (rank, got[rank])
    assert got[0][-1] == (0, 500) and got[1][-1] == (500, 1000)


def test_four_rank_tie_across_non_adjacent_ranks():
    """Four ranks: the minimum is attained on ranks 1 and 3 (not neighbours) -- the lower flat index, rank 1's,
    must win on every rank; then the same tie with the lower index on rank 3 (blocks need not be in rank
    order for the rule to hold)."""
    from covest_amd.grid import scan_min_pairs
    inf = math.inf
    cases = [
        [(5.0, 10), (2.0, 300), (4.0, 600), (2.0, 900)],
        [(5.0, 10), (2.0, 950), (inf, -1), (2.0, 900)],
        [(2.5, 10), (2.5, 300), (2.5, 600), (2.5, 900)],
        [(inf, -1), (inf, -1), (math.nan, 5), (3.0, 751)],
    ]
    want = [(2.0, 300), (2.0, 900), (2.5, 10), (3.0, 751)]
    assert [scan_min_pairs(c) for c in cases] == want  # the host statement of the rule agrees
    got = _run(4, cases)
    for rank in range(4):
        assert got[rank][:-1] == want, (rank, got[rank])
        assert got[rank][-1] == (250 * rank, 250 * (rank + 1))
