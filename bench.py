#!/usr/bin/env python3
"""bench.py -- grid-point log-likelihood evals/s of the likelihood grid search.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|c5|f2|f3] [--kernel auto]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: every
point of this rank's block of a dense parameter grid is evaluated on the GPU
(likelihood kernel), reduced to (min -LL, lowest index) on the GPU (arg-min
kernels), the 16-byte result is read back, and -- for N > 1 -- the global winner
is agreed with one RCCL all-gather of the ranks' 16-byte (min, index) pairs.  The histogram (model handle) and
the grid axes are resident in HBM before the timed region starts.

Workloads; the default is the one BASELINE.json's metric is
quoted on (repeat model, 10k-bin histogram):
  c3  RepeatsModel, H10k_rep.hist (10 000 keys), 32x32x16x1x16 grid (c,e,q1,q2=0.5,q)
  c2  BasicModel,   H10k_basic.hist (10 000 keys), 1000x1000 grid (c,e)
  c1  BasicModel,   H256.hist, 50x50 grid (the reference's CPU-runnable case)
  c5  (next row F1) canonical 21-mer histogram of synthetic reads, --kmer-gbp gigabases (default 10: config 5)
  f2  (next row F2) the `-sp 20` L-BFGS-B multi-start refinement of the repeats model on H10k_rep
  f3  (next row F3) histogram down-sampling (K-thin) of H10k_rep by a factor of 2
  og  covest.grid.optimize_grid end to end (the reference's consumer of batched evaluations): time-to-argmin
Weak scaling (default): with N ranks the c axis has N times as many values over the same
range and the flat index range is block-partitioned, one contiguous block of the
single-GPU size per rank.  --scaling strong: ONE fixed grid, c3 refined to c128 x e128 x q1 16 x q 16 =
4.2 M points (~20 ms on one GPU), cut into N contiguous blocks balanced by sum(T - 1)
(SURVEY.md 8(e)); the line then carries the ranks' kernel times (imbalance) and the cost of
the exchange.  With N > 1 the default (weak) invocation ALSO runs that strong-scaling grid in the same
process group and reports it under variants.strong (per-rank kernel times, imbalance, RCCL exchange cost,
speed-up over the one-GPU time recorded in profiles/): the driver issues one command per N, and the
north-star's ">= 6x at 8 GPUs" is a strong-scaling statement.
With N > 1 over RCCL the ranks' (min, index) pairs never visit the host: the
arg-min kernel's 16 bytes in HBM go into one all-gather, are scanned where they land, and 16
bytes are copied back.

Before the W warm-up steps a fixed, uncounted spin-up of 25 steps lets the device's clocks settle.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == dense fp64 matrix peak (AMD spec; 256 CU x 128 flop/clk x 2.4 GHz)
HBM_PEAK_GBPS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md


def load_hist(name):
    hist = {}
    with open(os.path.join(REPO, "tests", "golden", name + ".hist")) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            a, b = line.split()[:2]
            hist[int(a)] = int(b)
    return hist


def workload_tail(name):
    """The tail (covest/histogram.py:105-134: the mass trim_hist cuts off) a workload's model is built with: 0 for the
    SURVEY 8(d) configurations, the reference's own for the trimmed histograms (tests/golden/c{3,2}_trim.json)."""
    fixture = {"c3t": "c3_trim.json", "c2t": "c2_trim.json"}.get(name)
    if fixture is None:
        return 0
    with open(os.path.join(REPO, "tests", "golden", fixture)) as f:
        return json.load(f)["tail"]


def workload(name, n_ranks, scaling="weak"):
    """(model kind, histogram name, axes) -- SURVEY.md 8(d).  c3t / c2t: the same grids on the histograms the
    reference's process_histogram would hand the model -- trimmed by get_trim / trim_hist, with their tail
    (workload_tail): the shape every real CovEst run has."""
    if name in ("c3t", "c2t"):
        kind, _, axes = workload(name[:2], n_ranks, scaling)
        return kind, {"c3t": "H10k_rep_trim", "c2t": "H10k_basic_trim"}[name], axes
    if scaling == "strong":
        if name != "c3":
            raise SystemExit("--scaling strong is defined for the c3 workload")
        axes = [np.linspace(15.0, 30.0, 128), np.linspace(0.005, 0.08, 128),
                np.linspace(0.3, 0.95, 16), np.array([0.5]), np.linspace(0.05, 0.95, 16)]
        return "repeats", "H10k_rep", axes
    if name == "c3":
        axes = [np.linspace(15.0, 30.0, 32 * n_ranks), np.linspace(0.005, 0.08, 32),
                np.linspace(0.3, 0.95, 16), np.array([0.5]), np.linspace(0.05, 0.95, 16)]
        return "repeats", "H10k_rep", axes
    if name == "c2":
        return "basic", "H10k_basic", [np.linspace(2000.0, 6000.0, 1000 * n_ranks),
                                       np.linspace(0.001, 0.1, 1000)]
    if name == "c1":
        return "basic", "H256", [np.array([50 + i * 100 / (50 * n_ranks - 1) for i in range(50 * n_ranks)]),
                                 np.array([0.001 + i * 0.099 / 49 for i in range(50)])]
    raise SystemExit("unknown workload %r" % name)


def source_sha16():
    """sha256[:16] over the library's sources (covest_amd/csrc, include/): what ties profiles/pmc_traffic.json --
    counters of ANOTHER run -- to the library a bench line was measured on."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(REPO, "covest_amd", "csrc"), os.path.join(REPO, "include")):
        for name in sorted(os.listdir(d)):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def argmin_record(workload_name, gmin, gidx):
    """The step's arg-min; for the trimmed-histogram workloads beside the one the REFERENCE finds among the
    candidates of tests/golden/c{3,2}_trim.json (make_golden.py sections c3trim, c2trim)."""
    rec = {"min_negll": gmin, "flat_index": gidx}
    fixture = {"c3t": "c3_trim.json", "c2t": "c2_trim.json"}.get(workload_name)
    if fixture:
        with open(os.path.join(REPO, "tests", "golden", fixture)) as f:
            ref = json.load(f)["candidates"]
        rec["reference"] = {"min_negll": ref["reference_min_negll"], "flat_index": ref["reference_argmin_flat"]}
        rec["identical_index"] = gidx == ref["reference_argmin_flat"]
        rec["rel_err"] = abs(gmin - ref["reference_min_negll"]) / abs(ref["reference_min_negll"])
    return rec


def host_threads(cap=16):
    """Threads the CPU baseline may use: CPU affinity, limited by the cgroup quota
    (the GPU box grants a 16-CPU share whatever the affinity mask says)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def cpu_baseline(kind, hist, axes, budget_s, seed=20240521, tail=0):
    """The oracle (faithful restatement of the reference's O(j) long-double pmf
    product) timed on this host's cores on a bounded, seeded sample of the same
    grid.  Its cost per point is exactly proportional to (T-1) [x S x sum of
    keys], so the sample is drawn from the points whose predicted cost fits the
    budget and the rate is extrapolated to the whole grid BY WORK."""
    from oracle import covest_oracle as orc
    import itertools
    threads = host_threads()
    om = orc.OracleModel(kind, 21, 100, hist, tail, max_error=8)
    shape = [len(a) for a in axes]
    total = int(np.prod(shape))
    rng = np.random.default_rng(seed)
    if kind == "repeats":
        qs = np.array(list(itertools.product(*axes[2:5])))
        t_sub = np.array([orc.threshold_o(q1, q2, q, 1e-8, max(hist)) for q1, q2, q in qs])
        cost_sub = np.maximum(t_sub - 1, 0).astype(np.float64)
        mean_cost = float(cost_sub.mean())
    else:
        cost_sub = np.ones(1)
        mean_cost = 1.0
    n_sub = len(cost_sub)
    # calibrate seconds per unit of (T-1) on one cheapest point, single thread
    probe_sub = int(np.argmin(np.where(cost_sub > 0, cost_sub, np.inf)))
    probe_flat = probe_sub  # (c, e) index 0
    probe = [float(a[i]) for a, i in zip(axes, np.unravel_index(probe_flat, shape))]
    t0 = time.perf_counter()
    om.compute_loglikelihood_many(np.array([probe]), n_threads=1)
    sec_per_unit = (time.perf_counter() - t0) / max(cost_sub[probe_sub], 1.0)
    cap_units = max(1.0, budget_s / sec_per_unit)               # no single point longer than the budget
    want_units = budget_s * threads / sec_per_unit              # whole sample ~ budget on all threads
    order = rng.permutation(total)
    picked, units = [], 0.0
    for flat in order:
        u = float(cost_sub[flat % n_sub])
        if u <= 0 or u > cap_units:
            continue
        picked.append(int(flat))
        units += u
        if units >= want_units or len(picked) >= 4096:
            break
    if not picked:  # budget below the cheapest point: time that one point anyway
        picked, units = [probe_flat], float(max(cost_sub[probe_sub], 1.0))
    pts = np.array([[float(a[i]) for a, i in zip(axes, np.unravel_index(f, shape))] for f in picked])
    t0 = time.perf_counter()
    om.compute_loglikelihood_many(pts, n_threads=threads)
    wall = time.perf_counter() - t0
    evals_per_s = (units / wall) / mean_cost
    # second figure: the oracle's log-domain mode -- O(1) per pmf term and only the
    # bins that matter, i.e. the algorithm the GPU runs -- on a seeded sample of ~2 s
    n_fast, fast_wall = 16 * threads, 0.0
    while True:  # grow the seeded sample until it takes about a second
        n_fast = min(n_fast, total, 20000)
        fast_pts = np.array([[float(a[i]) for a, i in zip(axes, np.unravel_index(int(f), shape))]
                             for f in order[:n_fast]])
        t0 = time.perf_counter()
        om.compute_loglikelihood_many_fast(fast_pts, n_threads=threads)
        fast_wall = time.perf_counter() - t0
        if fast_wall >= 1.0 or n_fast >= min(total, 20000):
            break
        n_fast *= 4
    if kind == "repeats":
        fast_units = float(sum(cost_sub[int(f) % n_sub] for f in order[:n_fast]))
        fast_evals_per_s = (fast_units / fast_wall) / mean_cost
    else:
        fast_evals_per_s = n_fast / fast_wall
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": evals_per_s, "unit": "evals/s", "cores": threads, "kind": "port",
        "sample": "%d seeded grid points (sum(T-1)=%d, each <= %d) in %.1f s on %d threads of %s; "
                  "whole-grid rate extrapolated by work (grid mean T-1 = %.2f); oracle = faithful "
                  "O(j) long-double restatement of the reference" % (
                      len(picked), int(units), int(cap_units), wall, threads, model_name or "host CPU",
                      mean_cost),
        "sample_points": len(picked), "sample_wall_s": wall,
        "fast_mode": {"value": fast_evals_per_s, "unit": "evals/s", "cores": threads,
                      "sample": "%d seeded grid points in %.1f s, log-domain O(1)-per-term CPU mode of the oracle "
                                "(same algorithmic work as the GPU kernels)" % (n_fast, fast_wall)},
    }


def kmer_from_file(reads, n_reads, read_len, k, counts, cap_reads=10_000_000):
    """Config 5 'from a FASTA file': the first `cap_reads` reads of the same synthetic set written to a FASTA file on
    local disk (untimed), then timed end to end -- the C++ reader (parse + preprocess), covest_kmer_add (H2D copy +
    count) per 2^26-base batch -- the next batch parsed while the GPU counts this one --, the count-of-counts histogram.  Host-bound: reported beside `value`, never as it."""
    import tempfile
    from covest_amd import kmer_hist as kh
    n = min(n_reads, cap_reads)
    host = reads[:n * read_len].cpu().numpy().reshape(n, read_len)
    rec = np.empty((n, 3 + read_len + 1), dtype=np.uint8)
    rec[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
    rec[:, 3:3 + read_len] = host
    rec[:, -1] = 10
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "reads.fa")
        with open(path, "wb") as f:
            f.write(rec.tobytes())
        size = os.path.getsize(path)
        t0 = time.perf_counter()
        n_b = sum(b[3] for b in kh.ReadBatches(path, kh.NS_IGNORE))  # parse alone
        t_parse = time.perf_counter() - t0
        counts.clear()
        t0 = time.perf_counter()
        for bases, offs, m, n_bases in kh.ReadBatches(path, kh.NS_IGNORE):
            counts.add_packed(bases, offs, m, n_bases)
        hist = counts.histogram()
        wall = time.perf_counter() - t0
    windows = n * (read_len - k + 1)
    if n_b != n * read_len or sum(i * v for i, v in enumerate(hist)) != windows:
        raise SystemExit("k-mer histogram from the file inconsistent")
    return {"file_bytes": size, "reads": n, "parse_GBps": size / t_parse / 1e9, "end_to_end_s": wall,
            "end_to_end_kmers_per_s": windows / wall,
            "what": "FASTA on local disk -> C++ reader -> covest_kmer_add (host buffers, H2D inside) -> histogram"}


def bench_kmer(args):
    """Workload c5 (BASELINE.json config 5): canonical 21-mer abundance
    histogram of synthetic 100-bp reads (random genome, 1 % substitutions) resident in HBM.
    A step = clear the table, count every k-mer (hash + atomics), count-of-counts histogram."""
    import torch
    from covest_amd import kmer_hist as kh
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    k, read_len = 21, 100
    n_reads = int(args.kmer_gbp * 1e9) // read_len
    genome_len = max(1_000_000, n_reads * read_len // 40)  # 40x coverage
    gen = torch.Generator(device=dev)
    gen.manual_seed(20240601)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)  # A C G T
    genome = lut[torch.randint(0, 4, (genome_len,), device=dev, generator=gen)]
    reads = torch.empty(n_reads * read_len, dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    chunk = 2_000_000
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        starts = torch.randint(0, genome_len - read_len, (b - a,), device=dev, generator=gen)
        r = genome[starts[:, None] + ar[None, :]]
        err = torch.rand(r.shape, device=dev, generator=gen) < 0.01
        r = torch.where(err, lut[torch.randint(0, 4, r.shape, device=dev, generator=gen)], r)
        reads[a * read_len:b * read_len] = r.reshape(-1)
    del genome
    n_kmers = n_reads * (read_len - k + 1)
    # The step: count every k-mer of the resident reads and make the count-of-counts histogram -- main's loop,
    # bin/kmer_hist.py:77-89.  Round 3: through the PARTITIONED path (covest_kmer_count_reads_device, kmer_bulk.hip:
    # minimizer buckets of super-k-mer records in HBM, counted bucket by bucket in LDS); --kmer-path table: the
    # open-addressing table in HBM of rounds 1-2 (clear + one scattered load/atomic pair per occurrence + a sweep).
    expected_distinct = genome_len + int(0.01 * n_reads * read_len * k)
    use_table = args.kmer_path == "table"
    counts = kh.KmerCounts(k, canonical=True, min_slots=3 * expected_distinct if use_table else 1 << 20)
    stream = torch.cuda.current_stream().cuda_stream
    paths = set()

    def step(timers=None):
        if timers:
            timers[0].record()
        if use_table:
            counts.clear(stream)
            counts.add_device(reads.data_ptr(), n_reads, read_len, stream=stream, reserve=False)
            paths.add("table")
        else:
            paths.add(counts.count_reads_device(reads.data_ptr(), n_reads, read_len, stream=stream))
        if timers:
            timers[1].record()
        return counts.histogram()

    for _ in range(args.warmup):
        hist = step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        hist = step(evs[i])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_s = 1e-3 * sum(a.elapsed_time(b) for a, b in evs) / args.steps
    distinct = len(counts)
    counted = sum(i * v for i, v in enumerate(hist))  # every window lands in exactly one bin
    if counted != n_kmers or sum(hist) != distinct:
        raise SystemExit("k-mer histogram inconsistent: %d windows counted, %d expected" % (counted, n_kmers))
    partitioned = paths == {"partitioned"}
    partition = counts.partition_info() if partitioned else None
    # What bounds either path is the rate at which the memory side retires scattered atomics -- MEASURED here, on this
    # device, over an array the size of the one the run's atomics fall into (no constant from an earlier round): the
    # partitioned path pays one returning add per RECORD (a cursor per bucket), the table one load + add per OCCURRENCE.
    atomic_slots = partition["buckets"] if partitioned else counts.slots * 2
    atomics_per_s = kh.scatter_rate(atomic_slots, ops=1 << 28)
    if partitioned:
        records = partition["records"]
        scatter_s = 1e-3 * partition["ms"]["scatter"]
        # pass 1, the dominant kernel: every base once, and per record a 16-byte {first, end}, an 8-byte cursor add and
        # the 16-byte record
        alg_bytes = 1.0 * n_reads * read_len + 40.0 * records
        dominant = {"kernel": "kmer_tile_kernel<false> (pass 1: records to their buckets)", "kernel_ms_avg": 1e3 * scatter_s,
                    "algorithmic_bytes_per_launch": alg_bytes, "achieved": alg_bytes / scatter_s / 1e9,
                    "atomics": {"per_launch": records, "achieved_per_s": records / scatter_s,
                                "roof_per_s_measured": atomics_per_s, "frac": records / scatter_s / atomics_per_s,
                                "roof_slots": atomic_slots},
                    "passes_ms": partition["ms"],
                    "note": "bound by the memory side's scattered-atomic rate (one returning add per record, 64-byte "
                            "requests), not by bytes: `atomics.frac` is the fraction of the rate measured in this run"}
    else:
        alg_bytes = 1.0 * n_reads * read_len + 16.0 * n_kmers
        dominant = {"kernel": "kmer_count_fixed_kernel", "kernel_ms_avg": 1e3 * kernel_s,
                    "algorithmic_bytes_per_launch": alg_bytes, "achieved": alg_bytes / kernel_s / 1e9,
                    "atomics": {"per_launch": n_kmers, "achieved_per_s": n_kmers / kernel_s,
                                "roof_per_s_measured": atomics_per_s, "frac": n_kmers / kernel_s / atomics_per_s,
                                "roof_slots": atomic_slots},
                    "note": "scattered 8-byte load/CAS + 8-byte atomic add (same line) per k-mer occurrence"}
    out = {
        "metric": "k-mers/s, canonical k=21 abundance histogram (bin/kmer_hist.py path)",
        "value": n_kmers * args.steps / elapsed, "unit": "k-mers/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "C5: canonical 21-mers of %d synthetic 100-bp reads (%.2f Gbp, 40x of a random "
                               "genome, 1%% substitutions)" % (n_reads, n_reads * read_len / 1e9),
                   "kernel": "partitioned: kmer_tile (pass 0/1) + kmer_wave_count / kmer_bucket_count (pass 2)" if partitioned
                             else "kmer_count (table of %d slots)" % counts.slots,
                   "path": sorted(paths), "why_not_partitioned": getattr(counts, "why_not_partitioned", None),
                   "partition": partition, "device_ms_per_step": 1e3 * kernel_s,
                   "distinct_kmers": distinct, "windows_counted": counted,
                   "hist_head": hist[:6], "hist_peak": int(np.argmax(hist[5:]) + 5) if len(hist) > 6 else None},
        "roofline": dict({"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": dominant["achieved"] / HBM_PEAK_GBPS, "traffic": None}, **dominant),
    }
    if not args.compact:
        counts.clear(stream)  # (from_file counts through the table: an ordinary counter again)
        out["config"]["from_file"] = kmer_from_file(reads, n_reads, read_len, k, counts)
    if args.cpu_budget > 0:
        from oracle import kmer_oracle as ko
        n_s = 5000 if args.compact else 20000
        sample = [bytes(reads[i * read_len:(i + 1) * read_len].cpu().numpy()).decode() for i in range(n_s)]
        t0 = time.perf_counter()
        ko.histogram(sample, k, canonical=True)
        wall = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_s * (read_len - k + 1) / wall, "unit": "k-mers/s", "cores": 1,
                               "kind": "port", "sample": "first %d reads, numpy restatement of bin/kmer_hist.py "
                               "(oracle/kmer_oracle.py), %.1f s" % (n_s, wall)}
    print(json.dumps(out), flush=True)
    counts.close()


def bench_thin(args):
    """Workload f3: expected histogram after down-sampling H10k_rep (10 000 bins, every
    count 1..10000 present) by a factor of 2 -- covest/histogram.py:47-69 sample_histogram without its
    rounding.  A step = one K-thin launch pair (partial sums + ordered reduction), inputs resident in HBM."""
    import ctypes
    from covest_amd import _capi
    hist = load_hist("H10k_rep")
    factor = 2.0
    keys = np.array(list(hist.keys()), dtype=np.int32)
    counts = np.array([float(v) for v in hist.values()])
    top = int(keys.max())
    out = np.empty(top)
    ms = ctypes.c_double()
    ip, dp_ = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)

    def run(repeats):
        _capi.check(_capi.lib().covest_thin_histogram_timed(
            0, len(keys), keys.ctypes.data_as(ip), counts.ctypes.data_as(dp_), factor, top, out.ctypes.data_as(dp_),
            repeats, ctypes.byref(ms)), "covest_thin_histogram_timed")
        return ms.value

    run(max(args.warmup, 1))
    kernel_ms = run(args.steps)
    pairs = float(sum(int(i) for i in keys))  # (source i, target j <= i) pairs: one pmf value each
    flops = pairs * 29.0  # one exp at 25 flop (SURVEY 8(d) convention) + 4 for its argument and the accumulation
    occurrences = float((keys * counts).sum())
    if abs((np.arange(1, top + 1) * out).sum() / (occurrences / factor) - 1.0) > 1e-8:
        raise SystemExit("K-thin: occurrences not conserved")
    res = {
        "metric": "thinning pmf terms/s, histogram down-sampling (covest/histogram.py sample_histogram)",
        "value": pairs / (1e-3 * kernel_ms), "unit": "terms/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": kernel_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "F3: H10k_rep.hist, %d bins (counts 1..%d), factor 2" % (len(keys), top),
                   "kernel": "thin_partial_kernel + thin_sum_kernel", "terms": pairs},
        "roofline": {"bound": "mfma", "pipe": "fp64 VALU (one exp per term)", "achieved": flops / (1e-3 * kernel_ms) / 1e12,
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / (1e-3 * kernel_ms) / 1e12 / FP64_PEAK_TFLOPS,
                     "traffic": None, "kernel": "thin_partial_kernel", "kernel_ms_avg": kernel_ms,
                     "algorithmic_flops_per_launch": flops},
    }
    if args.cpu_budget > 0:
        from oracle import hist_oracle as ho
        n_s = len(keys)  # every bin: 5e7 terms, the reference's O(i) long-double recurrence per bin
        t0 = time.perf_counter()
        ho.thin_expected_c(keys[:n_s], counts[:n_s], factor, int(keys[:n_s].max()), faithful=True)
        wall = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": float(sum(int(i) for i in keys[:n_s])) / wall, "unit": "terms/s", "cores": 1,
                               "kind": "port", "sample": "bins 1..%d of the same histogram, C restatement of the reference's "
                               "loops (oracle_thin_expected, long double), %.2f s" % (n_s, wall)}
    print(json.dumps(res), flush=True)


def bench_refine(args):
    """Workload f2: the `-sp 20` multi-start refinement of covest/covest.py:41-70 --
    20 L-BFGS-B runs with finite-difference gradients -- on the repeats model and H10k_rep.hist.  A step = the
    whole multi-start.  Timed twice: the reference's call pattern on the same kernels (one likelihood per call,
    starts one after the other) and the batched lock-step path (one launch per round of all starts)."""
    import random
    from covest_amd import CoverageEstimator, RepeatsModel, initial_grid
    from covest_amd.estimator import _LockStep
    m = RepeatsModel(21, 100, load_hist("H10k_rep"), 0, max_error=8)
    est = CoverageEstimator(m)
    random.seed(20240521)
    starts = initial_grid([25.0, 0.02, 0.6, 0.5, 0.1], count=20, bounds=est.bounds)
    m.compute_loglikelihood(*starts[0])  # module load, handle creation
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        results = [est._optimize(s) for s in starts]  # the default schedule: one start after the other
    batched_s = (time.perf_counter() - t0) / steps
    evals = sum(6 * r.nfev for r in results)
    best = min(results, key=lambda r: r.fun)
    t0 = time.perf_counter()
    lock = _LockStep(est.negll_points, len(starts))
    locked = lock.map(est._optimize, starts)
    lock_s = time.perf_counter() - t0
    out = {
        "metric": "likelihood evaluations/s inside the 20-start L-BFGS-B refinement (repeat model, 10k-bin hist)",
        "value": evals / batched_s, "unit": "evals/s", "n_gpus": 1, "steps": steps, "warmup": 1,
        "ms_per_step": 1e3 * batched_s, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "F2: RepeatsModel k=21 r=100, H10k_rep.hist (981 bins evaluated), 20 starts "
                               "(initial_grid, seed 20240521), each gradient (6 points) one launch",
                   "kernel": "ll_factored in list mode: one workgroup per (point, key segment), chunks of 512 copy "
                             "numbers beyond that",
                   "evaluations": evals, "launches": sum(r.nfev for r in results),
                   "best_negll": float(best.fun), "best_x": [float(v) for v in best.x]},
        "lock_step": {"ms_per_step": 1e3 * lock_s, "launch_rounds": lock.rounds, "points_per_round": lock.points / lock.rounds,
                      "identical_results": all(np.array_equal(a.x, b.x) for a, b in zip(results, locked)),
                      "note": "all 20 starts as threads, one launch per round"},
    }
    if args.cpu_budget > 0:
        # the reference's pattern on the same GPU kernels: a subset of the starts, one evaluation per call
        sub = starts[:4]
        plain = CoverageEstimator(m, batched=False)
        t0 = time.perf_counter()
        seq = [plain._optimize(s) for s in sub]
        wall = time.perf_counter() - t0
        out["unbatched_gpu"] = {"value": sum(r.nfev for r in seq) / wall, "unit": "evals/s",  # (nfev counts every call here)
                                "sample": "the first 4 starts, one likelihood per launch, one start after the other "
                                          "(%.1f s); identical iterates: %s" % (
                                              wall, all(np.array_equal(a.x, b.x) for a, b in zip(seq, results)))}
        from oracle import covest_oracle as orc
        om = orc.OracleModel("repeats", 21, 100, load_hist("H10k_rep"), 0, max_error=8)
        pts = np.array([est._model_args(r.x) for r in results[:16]])
        t0 = time.perf_counter()
        om.compute_loglikelihood_many_fast(pts, n_threads=host_threads())
        wall = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(pts) / wall, "unit": "evals/s", "cores": host_threads(), "kind": "port",
                               "sample": "the 16 best end points, log-domain CPU mode of the oracle (the faithful O(j) "
                                         "restatement manages ~0.5 evals/s on this histogram), %.2f s" % wall}
    print(json.dumps(out), flush=True)


def strong_variant(cls, hist, args, world, rank, local_rank, xdev, on_device, stream):
    """N > 1: the strong-scaling answer in the SAME invocation as the weak one (the driver issues one command per N).
    ONE fixed grid -- c128 x e128 x q1 16 x q 16 = 4.2 M points, `workload("c3", 1, "strong")` -- cut into N
    contiguous flat-index blocks balanced by sum(T - 1) (covest/grid.py:63-64 maps the points over processes;
    SURVEY.md 8(e)); a step = every rank evaluates its block, reduces it on the device, one all-gather of the
    16-byte pairs.  Timed like the headline: barrier + synchronize on both sides, MAX over ranks.  The speed-up is
    quoted against the N = 1 time of the same grid recorded in profiles/ (a one-GPU bench line of this repository)."""
    import torch
    import torch.distributed as dist
    from covest_amd import DenseGrid
    from covest_amd.grid import distributed_argmin, partition_flat_range, repeats_cost_weights
    _, _, axes = workload("c3", 1, "strong")
    model = cls(21, 100, hist, 0, max_error=8, device=local_rank)
    total = int(np.prod([len(a) for a in axes]))
    bounds = partition_flat_range(total, world, repeats_cost_weights(model, axes))
    grid = DenseGrid(model, axes, (bounds[rank], bounds[rank + 1]))

    def exchange():
        if on_device:
            return distributed_argmin(None, None, pair=grid.argmin_pair_tensor(local_rank))
        lm, li = grid.argmin()
        return distributed_argmin(lm, li, device=xdev)

    def step():
        grid.evaluate(kernel=args.kernel, stream=stream)
        return exchange()

    for _ in range(3 + args.warmup):
        step()
    grid.profile(True)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gmin, gidx = step()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms, launches = grid.kernel_ms()
    grid.profile(False)
    mine = torch.tensor([kernel_ms / max(launches, 1)], dtype=torch.float64, device=xdev)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    per_rank = [float(v.item()) for v in every]
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        exchange()
    torch.cuda.synchronize()
    exchange_us = 1e6 * (time.perf_counter() - t1) / 20
    grid.close()
    model.close()
    out = {"what": "strong scaling: ONE fixed grid c128 x e128 x q1 16 x q2 1 x q 16 (%d points) cut into %d contiguous "
                   "blocks balanced by sum(T - 1)" % (total, world),
           "value": total * args.steps / elapsed, "unit": "evals/s", "ms_per_step": 1e3 * elapsed / args.steps,
           "grid_points": total, "argmin": {"min_negll": gmin, "flat_index": gidx},
           "per_rank_kernel_ms": per_rank, "kernel_imbalance": max(per_rank) / (sum(per_rank) / world),
           "exchange_us": exchange_us, "block_bounds": [int(b) for b in bounds],
           "exchange": "device-resident pair -> RCCL all-gather -> device scan -> 16-byte copy" if on_device
                       else "host pair -> all-gather (%s) -> scan" % args.backend}
    # the N = 1 time of the same grid, from this repository's own one-GPU bench line (newest round first)
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_c3_strong_1gpu.json")), reverse=True):
        name = os.path.basename(path)
        if os.path.exists(path):
            try:
                with open(path) as f:
                    one = json.loads(f.read().strip().splitlines()[-1])
                out["speedup_over_1gpu"] = {"value": one["ms_per_step"] / out["ms_per_step"],
                                            "one_gpu_ms_per_step": one["ms_per_step"], "source": "profiles/" + name,
                                            "note": "the N = 1 line was measured on another box of the same pool"}
                break
            except (OSError, ValueError, KeyError, IndexError):
                continue
    return out


def other_configs(args):
    """The default (driver-timed) run carries every single-GPU configuration of BASELINE.json, not only C3: config 2
    (basic model, 10^6 points), config 5 (k-mer histogram; 10 Gbp when >= 200 GB of HBM are free, else 1 Gbp), the
    reference's consumer of batched evaluations (optimize_grid) and configs 3 and 2 on the histograms the reference's own
    pipeline would hand the model (trimmed, with a tail: `tail`, `c2_tail`), each run as a CHILD process of this one -- its own
    model handles and HBM, its own JSON line -- and reported in compact form under variants.{c2,c5,og}.  The children
    run one after the other after this process's own timed region; their CPU baselines are capped at 4 s each."""
    import subprocess
    import torch
    free_b, _ = torch.cuda.mem_get_info()
    gbp = 10.0 if free_b >= 200e9 else 1.0
    runs = [("tail", ["--workload", "c3t", "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-budget", "0"]),
            ("c2", ["--workload", "c2", "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-budget", "4"]),
            ("c2_tail", ["--workload", "c2t", "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-budget", "0"]),
            ("c5", ["--workload", "c5", "--kmer-gbp", str(gbp), "--steps", "3", "--warmup", "1", "--cpu-budget", "1"]),
            ("og", ["--workload", "og", "--steps", "3"])]
    out = {}
    for name, extra in runs:
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--compact"] + extra, capture_output=True,
                               text=True, timeout=240)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                out[name] = {"error": "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:])}
                continue
            r = json.loads(line[-1])
        except (subprocess.TimeoutExpired, ValueError) as e:
            out[name] = {"error": repr(e)[:300]}
            continue
        roof = r.get("roofline", {})
        c = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
             "workload": r["config"]["workload"], "kernel": roof.get("kernel"), "kernel_ms_avg": roof.get("kernel_ms_avg"),
             "roofline": {"bound": roof.get("bound"), "frac": roof.get("frac"), "achieved": roof.get("achieved"),
                          "unit": roof.get("unit")},
             "cpu_baseline": {k: r["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")} if "cpu_baseline" in r else None,
             "wall_s": time.perf_counter() - t0}
        for k in ("frac_is", "executed_from", "algorithmic", "traffic"):
            if k in roof:
                c["roofline"][k] = roof[k]
        if "atomics" in roof:
            c["roofline"]["atomics_frac"] = roof["atomics"]["frac"]
            c["passes_ms"] = roof.get("passes_ms")
        if "argmin" in r:
            c["argmin"] = r["argmin"]
        for k in ("warmup", "spinup_steps"):
            if k in r:
                c[k] = r[k]
        if "cases" in r:
            c["cases"] = [{k: case[k] for k in ("histogram", "iterations", "points_evaluated", "time_to_argmin_s", "split_ms")}
                          for case in r["cases"]]
        out[name] = c
    return out


def bench_optimize_grid(args):
    """Workload og: the reference's actual consumer of batched evaluations -- covest.grid.optimize_grid
    (covest/grid.py:17-79), repeats model, free (c, e, q1, q2, q).  Case A: the reference's own 15-bin test
    histogram (its trace: 21 iterations of 3 888 - 7 776 points, 8.2 s on 8 processes).  Case B: the
    same search on H10k_rep (981 counted keys) -- out of the reference's reach (~0.6 core-seconds per copy number and
    point).  A step = one whole search: total time-to-argmin, with the split per iteration (grid handle + plan /
    kernels / read-back of the values for the host-side scan)."""
    from covest_amd import CoverageEstimator, RepeatsModel, optimize_grid
    cases = [("sim_c10_e0.05 (15 keys)", "sim_c10_e0.05", [10.0, 0.05, 0.65, 0.5, 0.5]),
             ("H10k_rep (981 counted keys)", "H10k_rep", [25.0, 0.02, 0.6, 0.5, 0.1])]
    out_cases = []
    for label, hname, guess in cases:
        m = RepeatsModel(21, 100, load_hist(hname), 0, max_error=8)
        est = CoverageEstimator(m)
        est.likelihood_f(guess)  # module load, handle creation
        walls, last = [], None
        for rep in range(max(1, min(args.steps, 5)) + 1):
            est.timings = []
            t0 = time.perf_counter()
            og_trace = []
            res = optimize_grid(est.likelihood_f, list(guess), bounds=est.bounds, trace=og_trace)
            wall = time.perf_counter() - t0
            if rep:  # the first search warms everything up
                walls.append(wall)
            last = (res, est.timings, og_trace)
        res, timings, trace = last
        pts = sum(t["points"] for t in timings)
        out_cases.append({
            "histogram": label, "iterations": len(trace), "points_evaluated": pts,
            "time_to_argmin_s": float(np.median(walls)), "evals_per_s": pts / float(np.median(walls)),
            "split_ms": {"grid_handle_and_plan": 1e3 * sum(t["create_s"] for t in timings),
                         "kernels": 1e3 * sum(t["eval_s"] for t in timings),
                         "read_back": 1e3 * sum(t["readback_s"] for t in timings)},
            "kernels_used": sorted(set(t["kernel"] for t in timings)),
            "largest_grid": max(t["points"] for t in timings),
            "result": [float(v) for v in res], "min_negll": float(trace[-1]["value"])})
        m.close()
    a = out_cases[0]
    print(json.dumps({
        "metric": "optimize_grid time-to-argmin (repeats model, 5 free parameters)", "value": a["evals_per_s"],
        "unit": "evals/s", "n_gpus": 1, "steps": len(walls), "warmup": 1, "ms_per_step": 1e3 * a["time_to_argmin_s"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "OG: covest.grid.optimize_grid over CoverageEstimator.likelihood_f, each iteration's grid "
                               "one batched evaluation"},
        "cases": out_cases,
        "reference": {"sim_c10_e0.05": "21 iterations, 8.2 s on 8 processes (SURVEY.md 3.2, measured in the survey container)"}}),
        flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c3t", "c2t", "c5", "f2", "f3", "og"])
    ap.add_argument("--kmer-gbp", type=float, default=10.0,
                    help="c5: gigabases of synthetic reads (BASELINE.json config 5: 10; ~65 GB of HBM)")
    ap.add_argument("--kmer-path", default="partitioned", choices=["partitioned", "table"],
                    help="c5: the partitioned path (round 3) or the table in HBM of rounds 1-2")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: one fixed 4.2 M-point c3 grid cut into N blocks balanced by sum(T - 1)")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (nccl == RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--compact", action="store_true",
                    help="(set by the default run for its child runs) leave out the legs that only feed profiles/: c5's "
                         "from-file leg, c3's tail variant and the children themselves")
    ap.add_argument("--no-variants", action="store_true",
                    help="c3, one GPU: do not run the other single-GPU configurations (c2, c5, og) as child processes")
    args = ap.parse_args()
    if args.workload == "c5":
        return bench_kmer(args)
    if args.workload == "f3":
        return bench_thin(args)
    if args.workload == "f2":
        return bench_refine(args)
    if args.workload == "og":
        return bench_optimize_grid(args)

    import torch
    import torch.distributed as dist
    from covest_amd import BasicModel, DenseGrid, RepeatsModel
    from covest_amd.grid import distributed_argmin, partition_flat_range, repeats_cost_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d processes" % (args.gpus, args.gpus))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    xdev = device if args.backend == "nccl" else None  # where the 16-byte exchange lives

    kind, hist_name, axes = workload(args.workload, world, args.scaling)
    tail = workload_tail(args.workload)
    hist = load_hist(hist_name)
    cls = BasicModel if kind == "basic" else RepeatsModel
    on_device = world > 1 and args.backend == "nccl"  # the exchange consumes the arg-min kernel's output in HBM

    # ---- time-to-argmin: host axes + histogram -> global (min, index) on the host ----
    # (the HIP runtime creates its queue and staging buffers on a process's first device operation, 80 ms that are
    # not this library's: one torch fill kernel and one 8-byte copy before the clock starts.  What is left in the
    # first call is the library's own first use: its code objects, its first allocations, the ln j! table.)
    torch.zeros(8, device=device).cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model = cls(21, 100, hist, tail, max_error=8, device=local_rank)
    shape = [len(a) for a in axes]
    total = int(np.prod(shape))
    # contiguous flat-index blocks, one per rank, balanced by sum(T - 1) for the repeats model (covest_amd.grid):
    # T depends on the (q1, q2, q) sub-index only, so the weights are known on the host up front
    weights = repeats_cost_weights(model, axes) if kind == "repeats" and world > 1 else None
    bounds = partition_flat_range(total, world, weights)
    block = (bounds[rank], bounds[rank + 1])
    grid = DenseGrid(model, axes, block)
    stream = torch.cuda.current_stream().cuda_stream

    def exchange():
        if on_device:
            return distributed_argmin(None, None, pair=grid.argmin_pair_tensor(local_rank))
        lm, li = grid.argmin()
        return distributed_argmin(lm, li, device=xdev)

    def step():
        grid.evaluate(kernel=args.kernel, stream=stream)
        return exchange()

    gmin, gidx = step()
    time_to_argmin_first = time.perf_counter() - t0   # includes HIP module load on first use

    # W untimed warm-up steps as asked, preceded by a fixed spin-up that is not counted either: the device
    # needs a few milliseconds of work before its clocks settle (at --warmup 1 the first timed steps ran 13 %
    # slower than steady state).  The timed steps follow the warm-up DIRECTLY.
    # (round 5: the spin-up is a stretch of WORK, 20 ms of steps and 25 steps at least -- `r04_clock_ramp.txt`: the clocks
    # settle 14 ms in; 25 steps of the basic-model grid are 4.5 ms, and its child line of the default run read 0.178 to
    # 0.185 ms a step from run to run where 400 steps read 0.170.  The count is in the line: spinup_steps)
    spinup_steps = int(os.environ.get("COVEST_BENCH_SPINUP", "25"))
    if "COVEST_BENCH_SPINUP" not in os.environ and world == 1:  # (N > 1: a step holds a collective -- the same count on every rank)
        t_probe = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        per_step = (time.perf_counter() - t_probe) / 5
        spinup_steps = max(25, min(400, int(0.020 / max(per_step, 1e-6)) + 1)) + 5
        for _ in range(spinup_steps - 5 + args.warmup):
            step()
    else:
        for _ in range(spinup_steps + args.warmup):
            step()
    grid.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gmin, gidx = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = grid.kernel_ms()
    grid.profile(False)
    terms, flops, kernel_name = grid.work()
    # (measured AFTER the timed region since round 4: between the warm-up and the timed steps its ten searches, with
    # their host-side gaps, let the device's clocks sag, and the 20 timed steps then ran 3-5 % under the sustained rate
    # -- profiles/r04_clock_ramp.txt: 0.718 ms a step for the first 20 steps from idle, 0.685 from 14 ms on)
    # warm time-to-argmin (library and context warm; model + grid handles re-created): host axes + histogram ->
    # global (min, index) on the host.  Twice: the histogram handed over as the reference's dict {j: count}
    # (covest/models.py:26 -- walking a 10 000-key dict costs 0.3 ms of host time) and as (keys, counts) arrays,
    # which covest_amd's models accept as well; median of 5 each.
    hist_arrays = (np.fromiter(hist.keys(), dtype=np.int32, count=len(hist)),
                   np.fromiter(hist.values(), dtype=np.float64, count=len(hist)))

    def warm_search(h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model2 = cls(21, 100, h, tail, max_error=8, device=local_rank)
        t1 = time.perf_counter()
        model2.handle
        t2 = time.perf_counter()
        grid2 = DenseGrid(model2, axes, block)
        t3 = time.perf_counter()
        grid2.evaluate(kernel=args.kernel, stream=stream)
        if on_device:
            distributed_argmin(None, None, pair=grid2.argmin_pair_tensor(local_rank))
        else:
            lm, li = grid2.argmin()
            distributed_argmin(lm, li, device=xdev)
        t4 = time.perf_counter()
        grid2.close()
        model2.close()
        return t4 - t0, {"model_handle": 1e3 * (t2 - t0), "grid_handle_and_plan": 1e3 * (t3 - t2),
                         "launch_kernels_readback": 1e3 * (t4 - t3)}

    runs_dict = sorted((warm_search(hist) for _ in range(5)), key=lambda r: r[0])
    runs_arr = sorted((warm_search(hist_arrays) for _ in range(5)), key=lambda r: r[0])
    time_to_argmin_warm = runs_arr[2][0]
    time_to_argmin_split = runs_arr[2][1]
    time_to_argmin_warm_dict = runs_dict[2][0]

    # N > 1: every rank's likelihood-kernel time (imbalance of the partition) and the price of the exchange alone
    per_rank_kernel_ms, exchange_us = None, None
    if world > 1:
        mine = torch.tensor([kernel_ms / max(launches, 1)], dtype=torch.float64, device=xdev)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_kernel_ms = [float(t.item()) for t in every]
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            exchange()  # (the pairs of the last step are still in place)
        torch.cuda.synchronize()
        exchange_us = 1e6 * (time.perf_counter() - t1) / 20
    variants = None
    if rank == 0 and world == 1 and args.workload == "c3" and args.scaling == "weak" and not args.compact \
            and not args.no_variants:
        variants = other_configs(args)
    if world > 1 and args.workload == "c3" and args.scaling == "weak":  # (every rank takes part)
        variants = {"strong": strong_variant(cls, hist, args, world, rank, local_rank, xdev, on_device, stream)}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total * args.steps / elapsed
        avg_kernel_s = 1e-3 * kernel_ms / max(launches, 1)
        achieved_tflops = flops / avg_kernel_s / 1e12
        n_local = block[1] - block[0]
        # algorithmic HBM bytes of one launch: axes in, 8 B/point LL out, histogram once
        alg_bytes = 8.0 * sum(shape) + 8.0 * n_local + 24.0 * model.bins_evaluated
        traffic, executed, profile_sha = None, None, None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    doc = json.load(f)
                rec = doc.get(args.workload, {})
                traffic = rec.get(kernel_name)
                executed = rec.get(kernel_name + "_executed_flops")
                profile_sha = rec.get("_source_sha16", doc.get("_source_sha16"))
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": "grid-point log-likelihood evals/s, repeat model, 10k-bin hist"
                      if args.workload == "c3" else "grid-point log-likelihood evals/s (%s)" % args.workload,
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "spinup_steps": spinup_steps, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": {"c3": "C3: RepeatsModel k=21 r=100 S=8, H10k_rep.hist (10000 keys, %d evaluated: tail=0), "
                                   "grid c%dxe%dxq1 16xq2 1xq 16" % (model.bins_evaluated, shape[0], shape[1]),
                             "c2": "C2: BasicModel k=21 r=100 S=8, H10k_basic.hist (10000 keys, %d evaluated: tail=0), "
                                   "grid c%dxe1000" % (model.bins_evaluated, shape[0]),
                             "c3t": "C3 grid on H10k_rep as the reference trims it (covest/histogram.py:105-134): %d keys, all "
                                    "evaluated, tail = %d; grid c%dxe%dxq1 16xq2 1xq 16" % (model.bins_evaluated, tail, shape[0], shape[1]),
                             "c2t": "C2 grid on H10k_basic as the reference trims it: %d keys, all evaluated, tail = %d; "
                                    "grid c%dxe1000" % (model.bins_evaluated, tail, shape[0]),
                             "c1": "C1: BasicModel k=21 r=100 S=8, H256.hist, grid c%dxe50" % shape[0]}[args.workload],
                "grid_points": total, "points_per_gpu": n_local, "kernel": kernel_name,
                "library_source_sha16": source_sha16(),
                "partition": "contiguous flat-index block per GPU balanced by sum(T-1), one RCCL all-gather of 16-byte "
                             "(min, index) pairs per step (taken from the arg-min kernel's output in HBM, scanned on the "
                             "device, 16 bytes copied back)",
            },
            "argmin": argmin_record(args.workload, gmin, gidx),
            "time_to_argmin_ms": {"first_call_incl_module_load": 1e3 * time_to_argmin_first,
                                  "warm": 1e3 * time_to_argmin_warm, "warm_split": time_to_argmin_split,
                                  "warm_histogram_as_dict": 1e3 * time_to_argmin_warm_dict,
                                  "what": "host axes + histogram (arrays; or the reference's dict) -> model handle, grid "
                                          "handle + plan, kernels, 16-byte read-back; median of 5"},
            "roofline": {
                "bound": "mfma", "pipe": "fp64 VALU + fp64 MFMA: ONE shared fp64 datapath (tools/microbench_mix.hip); "
                                         "78.6 TFLOP/s is both the fp64 vector and the dense fp64 MFMA peak of MI355X",
                "achieved": achieved_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / FP64_PEAK_TFLOPS,
                "frac_is": "algorithmic flops (covest_grid_work) / this run's kernel time / peak: no PMC profile of this workload",
                "traffic": traffic,
                "traffic_from": None if traffic is None else "profiles/pmc_traffic.json: PMC passes of the same kernel "
                                "and workload under rocprofv3 (tools/pmc_profile.sh), 2 x FETCH_SIZE + WRITE_SIZE per "
                                "launch; NOT counted in this run",
                "kernel": kernel_name, "kernel_ms_avg": 1e3 * avg_kernel_s, "launches": launches,
                "algorithmic_flops_per_launch": flops, "pmf_terms_per_launch": terms,
                "hbm": {"achieved": alg_bytes / avg_kernel_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": alg_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_launch": alg_bytes},
            },
        }
        if executed and world == 1 and args.scaling == "weak":
            # `frac` is on the fp64 flops the kernel EXECUTES (PMC: 64 x (2 FMA + MUL + ADD) + 2048 per MFMA, from a
            # profiled run of this workload: profiles/pmc_traffic.json) over THIS run's kernel time.  Both kernels skip
            # work their "algorithmic" counts still hold -- K-basic's closed form leaves most of SURVEY's pmf terms
            # unevaluated, K-factored drops the units whose sums are -inf -- so a count of what would have to be done
            # without those shortcuts says nothing about how busy the fp64 pipe is; those counts stay beside it, labelled.
            # The profile is tied to the sources it was taken on (_source_sha16): on other sources it is marked stale.
            here = source_sha16()
            out["roofline"]["algorithmic"] = {
                "flops_per_launch": flops, "achieved": achieved_tflops, "frac": achieved_tflops / FP64_PEAK_TFLOPS,
                "note": ("SURVEY 8(d)'s unit (4 flop per pmf term of every class and key + 25 per log): work the kernel "
                         "skips by its closed form; not a utilisation") if kernel_name == "ll_basic" else
                        "the factored formulation's own count (covest_grid_work), INCLUDING the units the kernel skips "
                        "once their sums are -inf; not a utilisation"}
            out["roofline"]["executed_flops_per_launch"] = executed
            out["roofline"]["achieved"] = executed / avg_kernel_s / 1e12
            out["roofline"]["frac"] = executed / avg_kernel_s / 1e12 / FP64_PEAK_TFLOPS
            out["roofline"]["frac_is"] = "executed fp64 flops (PMC profile of this workload) / this run's kernel time / peak"
            out["roofline"]["executed_from"] = {"file": "profiles/pmc_traffic.json", "source_sha16": profile_sha,
                                                "this_library_source_sha16": here,
                                                "stale": profile_sha is not None and profile_sha != here}
        if kernel_name == "ll_factored" and world == 1 and model.tail == 0:
            # The same kernel time on ROUND 1's flop count of K-factored (one multiply-add per (key, column, o < T)
            # in the contraction; since round 2 the steps below a q-tile's smallest cut-off are summed once per key --
            # DESIGN.md 4): for comparison with the 0.37 of round 1 only, `frac` above is the honest one.
            import itertools
            q123 = np.array(list(itertools.product(*axes[2:5])))
            t_q = model.get_hist_threshold_values(q123).astype(np.float64)
            bins, n_ce, max_o = float(model.bins_evaluated), float(shape[0] * shape[1]), float(t_q.max() - 1)
            flops_r1 = n_ce * (bins * 8 * max_o * 2.0 + bins * float((t_q - 1).sum()) * 2.0 + bins * len(t_q) * 25.0
                               + 25.0 * 8 * max_o)
            out["roofline"]["round1_flop_count"] = {"algorithmic_flops_per_launch": flops_r1,
                                                    "frac": flops_r1 / avg_kernel_s / 1e12 / FP64_PEAK_TFLOPS}
        if per_rank_kernel_ms is not None:
            out["multi_gpu"] = {"per_rank_kernel_ms": per_rank_kernel_ms,
                                "kernel_imbalance": max(per_rank_kernel_ms) / (sum(per_rank_kernel_ms) / world),
                                "exchange_us": exchange_us, "block_bounds": [int(b) for b in bounds],
                                "exchange": "device-resident pair -> all-gather -> device scan -> 16-byte copy"
                                            if on_device else "host pair -> all-gather (%s) -> scan" % args.backend}
        if variants is not None:
            out["variants"] = variants
            if "strong" in variants:
                out["config"]["scaling_note"] = (
                    "`value` is the WEAK-scaling step (the c axis grows N-fold: per-GPU work fixed, what the driver's "
                    "efficiency column measures); variants.strong is the north-star question -- one fixed 4.2 M-point "
                    "grid divided among the N GPUs -- answered in the same run")
        if world == 1 and args.cpu_budget > 0:
            out["cpu_baseline"] = cpu_baseline(kind, hist, axes, args.cpu_budget, tail=tail)
        print(json.dumps(out), flush=True)

    grid.close()
    model.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
