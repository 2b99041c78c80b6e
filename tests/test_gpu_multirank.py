"""Config 4's machinery executed for real on the one GPU a test box has (`-m gpu`).

The reference fans ONE grid out to worker processes (covest/grid.py:63-64) from a single `covest` process
(covest/covest.py:86-89).  Two counterparts, both run here on the REAL C3 grid (262 144 points, H10k_rep) and required
to agree with the reference-judged arg-min of tests/golden/c3_argmin.json (flat 165 489):

* several RANKS, one process each, every rank evaluating its sum(T - 1)-balanced block through DenseGrid and the ranks
  agreeing through ONE all-gather of 16-byte pairs (covest_amd.grid.dense_grid_argmin) -- 2 and 4 fresh child
  processes that share device 0 and exchange over gloo (a box has one card; the pool allows six processes on it).
  What differs on an 8-GPU node is the transport (RCCL) and the device ordinal, not this code;
* ONE process driving several devices (dense_grid_argmin(devices=[...])): a model and a grid handle per device, host
  scan of the pairs -- with the same ordinal listed two and four times.

No scaling curve is measured here: the blocks share one card.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, load_hist, rel_err

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK_SCRIPT = r'''
import json, os, sys
import torch                      # FIRST: one HIP runtime per process (INTEGRATION.md)
import torch.distributed as dist
sys.path.insert(0, os.environ["COVEST_REPO"]); sys.path.insert(0, os.path.join(os.environ["COVEST_REPO"], "tests"))
import numpy as np
from conftest import load_hist
from covest_amd import RepeatsModel, dense_grid_argmin
from covest_amd.grid import partition_flat_range, repeats_cost_weights
dist.init_process_group("gloo")   # RANK / WORLD_SIZE / MASTER_* from the environment
rank, world = dist.get_rank(), dist.get_world_size()
axes = [np.linspace(15.0, 30.0, 32), np.linspace(0.005, 0.08, 32), np.linspace(0.3, 0.95, 16), np.array([0.5]),
        np.linspace(0.05, 0.95, 16)]
m = RepeatsModel(21, 100, load_hist("H10k_rep"), 0, max_error=8, device=0)
best = dense_grid_argmin(m, axes)
bounds = partition_flat_range(262144, world, repeats_cost_weights(m, axes))
print("RESULT " + json.dumps({"rank": rank, "min": best[0], "index": best[1], "params": list(best[2]),
                               "block": [int(bounds[rank]), int(bounds[rank + 1])]}), flush=True)
m.close()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_share_one_gpu_and_agree_over_gloo(hip_lib, world):
    fix = load_golden("c3_argmin.json")["tail0"]
    port = _free_port()
    procs = []
    for rank in range(world):  # fresh processes, started before anything in them touches the GPU
        env = dict(os.environ, COVEST_REPO=REPO, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _RANK_SCRIPT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    results = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-2000:] + err[-4000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")]
        assert line, out[-2000:] + err[-2000:]
        results.append(json.loads(line[-1][7:]))
    assert sorted(r["rank"] for r in results) == list(range(world))
    blocks = sorted(tuple(r["block"]) for r in results)
    assert blocks[0][0] == 0 and blocks[-1][1] == 262144 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    for r in results:  # every rank holds the same, reference-judged answer
        assert r["index"] == fix["reference_argmin_flat"] == 165489
        assert rel_err(r["min"], fix["reference_min_negll"]) <= 1e-9
        assert r["min"] == results[0]["min"] and r["params"] == results[0]["params"]


def test_one_process_drives_several_devices(hip_lib):
    from covest_amd import DenseGrid, RepeatsModel, dense_grid_argmin
    from covest_amd.grid import DeviceBlocks
    fix = load_golden("c3_argmin.json")["tail0"]
    axes = [np.linspace(15.0, 30.0, 32), np.linspace(0.005, 0.08, 32), np.linspace(0.3, 0.95, 16), np.array([0.5]),
            np.linspace(0.05, 0.95, 16)]
    m = RepeatsModel(21, 100, load_hist("H10k_rep"), 0, max_error=8, device=0)
    whole = DenseGrid(m, axes)
    whole.evaluate()
    best = whole.argmin()
    ll = whole.loglikelihoods()
    assert best[1] == fix["reference_argmin_flat"]
    for devices in ([0], [0, 0], [0, 0, 0, 0]):
        got = dense_grid_argmin(m, axes, devices=devices)
        assert (got[0], got[1]) == best and list(got[2]) == list(whole.point(best[1])), (devices, got, best)
    # the blocks' values are the whole grid's, and a model handed over on another ordinal is copied, not moved
    blocks = DeviceBlocks(m, axes, [0, 0, 0])
    blocks.evaluate()
    assert blocks.argmin() == best
    assert np.array_equal(np.concatenate([g.loglikelihoods() for g in blocks.grids]), ll, equal_nan=True)
    assert all(g.model is m for g in blocks.grids) and blocks.bounds[0] == 0 and blocks.bounds[-1] == whole.total
    blocks.close()
    twin = m.on_device(0)
    assert twin is m
    other = m.on_device(7)  # (plain data only: no handle is created until it is used)
    assert other is not m and other.device == 7 and other._handle is None and other.tail == m.tail and other.k == m.k
    # NaN blocks, ties across blocks: the host scan keeps the lowest index
    small = [np.array([10.0, 10.0]), np.array([0.05]), np.array([0.8]), np.array([0.5]), np.array([0.3, 0.3, float("nan")])]
    sm = RepeatsModel(21, 100, load_hist("sim_c10_e0.05"), 0, max_error=8, device=0)
    one = dense_grid_argmin(sm, small, devices=[0])
    assert one[1] == 0
    for devices in ([0, 0], [0, 0, 0], [0, 0, 0, 0, 0, 0]):
        assert dense_grid_argmin(sm, small, devices=devices)[:2] == one[:2]
    sm.close()
    whole.close()
    m.close()
