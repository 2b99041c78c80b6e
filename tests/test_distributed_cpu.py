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


def test_eight_rank_exchange_and_balanced_partition():
    """World size 8 -- the node the north star names (8 x MI355X): the exchange over gloo with the minimum on several
    ranks, NaN / empty blocks among them, every rank agreeing; and the C3-shaped partition cut for 8 ranks by the
    cost weights (sum(T - 1) per (q1, q2, q) sub-index) is contiguous, complete and balanced to 2 %."""
    import numpy as np
    from covest_amd.grid import partition_flat_range, scan_min_pairs
    inf, nan = math.inf, math.nan
    cases = [
        [(9.0, 10), (8.0, 130), (7.0, 260), (6.0, 380), (5.0, 510), (4.0, 640), (3.0, 760), (2.0, 880)],
        [(2.0, 10), (2.0, 130), (2.0, 260), (2.0, 380), (2.0, 510), (2.0, 640), (2.0, 760), (2.0, 880)],
        [(inf, -1), (nan, 130), (3.0, 260), (inf, -1), (3.0, 500), (nan, 640), (inf, -1), (3.5, 880)],
        [(inf, -1)] * 8,
        [(5.0, 124), (5.0, 249), (-inf, 300), (5.0, 499), (-inf, 500), (1.0, 700), (0.0, 800), (nan, 900)],
    ]
    want = [(2.0, 880), (2.0, 10), (3.0, 260), (inf, -1), (-inf, 300)]
    assert [scan_min_pairs(c) for c in cases] == want
    got = _run(8, cases)
    for rank in range(8):
        assert got[rank][:-1] == want, (rank, got[rank])
        assert got[rank][-1] == (125 * rank, 125 * (rank + 1))
    # the partition of a repeats grid: weights per (q1, q2, q) sub-index, repeated for every (c, e)
    rng = np.random.default_rng(8)
    w = np.sort(rng.integers(8, 285, size=256)).astype(np.float64)[::-1].copy()
    total = 128 * 128 * 256
    bounds = partition_flat_range(total, 8, w)
    assert bounds[0] == 0 and bounds[-1] == total and all(a < b for a, b in zip(bounds[:-1], bounds[1:]))
    csum = np.concatenate(([0.0], np.cumsum(np.tile(w, 64))))  # cost of 64 (c, e) pairs: the pattern repeats
    per = len(w) * 64

    def cost(a, b):
        full, ra, rb = (b // per) - (a // per), a % per, b % per
        return full * csum[-1] + csum[rb] - csum[ra]

    costs = [cost(a, b) for a, b in zip(bounds[:-1], bounds[1:])]
    assert max(costs) <= 1.02 * (sum(costs) / 8), costs
