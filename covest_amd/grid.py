"""Grid search on the GPU: the counterpart of covest/grid.py.

Two entry points:

* `DenseGrid` -- a rectangular grid given by its axes (itertools.product order,
  last axis fastest): evaluates every point on this process's GPU block and
  reduces to (min -LL, lowest flat index).  With torch.distributed initialised
  (one process per GPU, RCCL over xGMI) the flat index range is block-partitioned
  over the ranks and ONE tiny exchange picks the global winner.

* `optimize_grid` -- same signature and semantics as covest/grid.py:17-79 (the
  iterative multiplicative local grid).  When `fn` is the `likelihood_f` of a
  `CoverageEstimator` over a GPU-backed model, each iteration's whole grid is one
  batched launch instead of a `Pool.map` of pickled calls; any other callable is
  evaluated point by point.

Selection rule everywhere: the sequential scan of covest/grid.py:65-70 --
strict <, first index wins ties, NaN never wins.
"""
import ctypes
import itertools
import math
import random

import numpy as np

from . import _capi, constants

_DP = ctypes.POINTER(ctypes.c_double)


# --------------------------------------------------------------------------- partition
def partition_flat_range(total, world_size, weights=None, period=None):
    """Contiguous blocks [b_r, b_{r+1}) of range(total), one per rank.

    Without weights the blocks differ by at most one point.  With `weights`
    (cost of the point at flat index i is weights[i % period], e.g. T-1 of the
    repeats model, which depends only on the (q1,q2,q) sub-index) the cuts are
    placed so every block carries about the same total cost.
    Returns a list of world_size+1 non-decreasing bounds, bounds[0]=0, bounds[-1]=total.
    """
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if weights is None or total == 0:
        return [(total * r) // world_size for r in range(world_size + 1)]
    w = np.asarray(weights, dtype=np.float64)
    period = len(w) if period is None else period
    assert period == len(w) and total % period == 0
    prefix = np.concatenate(([0.0], np.cumsum(w)))
    per_period = prefix[-1]
    n_periods = total // period
    whole = per_period * n_periods
    bounds = [0]
    for r in range(1, world_size):
        target = whole * r / world_size
        if per_period <= 0:
            cut = (total * r) // world_size
        else:
            full = min(int(target // per_period), n_periods)
            rem = target - full * per_period
            inner = int(np.searchsorted(prefix, rem, side="left")) if full < n_periods else 0
            cut = min(total, full * period + min(inner, period))
        bounds.append(max(cut, bounds[-1]))
    bounds.append(total)
    return bounds


def scan_min_pairs(pairs):
    """The selection scan of covest/grid.py:65-70 over the ranks' (min -LL, lowest flat index) pairs, taken in
    rank order (= flat-index order of the blocks): strict <, the lowest index wins a tie, NaN never wins, an
    index < 0 means "this block had nothing below +inf".  Returns (min, index), (inf, -1) if nobody has one.
    The host-side statement of the rule; `_scan_pairs_tensor` is the same on a tensor."""
    best, arg = math.inf, -1
    for val, idx in pairs:
        val, idx = float(val), int(idx)
        if idx < 0 or val != val or val == math.inf:
            continue
        if val < best or (val == best and idx < arg):
            best, arg = val, idx
    return best, arg


def _scan_pairs_tensor(gathered):
    """scan_min_pairs on a (world, 2) float64 tensor, wherever it lives (HBM after an RCCL all-gather, host memory
    under gloo): returns a 2-element tensor {min, index} -- index -1 if nothing is below +inf."""
    import torch
    vals, idx = gathered[:, 0], gathered[:, 1]
    inf = torch.full_like(vals, math.inf)
    vals = torch.where((idx < 0) | torch.isnan(vals), inf, vals)
    best = vals.min()
    arg = torch.where(vals == best, idx, inf).min()  # the lowest index among the holders of the minimum
    arg = torch.where(best < math.inf, arg, torch.full_like(arg, -1.0))
    return torch.stack([best, arg])


def distributed_argmin(local_min, local_idx, group=None, device=None, pair=None):
    """Global (min, lowest flat index) from every rank's local pair.

    RCCL has no MINLOC for fp64.  ONE collective: every rank contributes the 16-byte pair
    (value, index as an exactly representable double -- flat indices stay below 2^53) to an all-gather
    and scans the N pairs itself: strict <, lowest index on ties, as the scan of covest/grid.py:65-70
    would over the concatenated blocks.  NaN is mapped to +inf first (a NaN never wins, :67).
    `pair`: the rank's pair already in a 2-element float64 tensor (DenseGrid.argmin_pair_tensor: the arg-min
    kernel's own output in HBM) -- then nothing visits the host between the likelihood kernel and the
    collective; the gathered pairs are scanned where they land and 16 bytes are copied back, once.
    Works with any backend (nccl == RCCL on GPUs, gloo on CPU for tests).
    Without an initialised process group this is the identity.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if pair is not None:
            v, i = pair.cpu().tolist()
            return v, int(i)
        return local_min, local_idx
    world = dist.get_world_size(group)
    if pair is None:
        v = float(local_min)
        if v != v or local_idx < 0:
            v, local_idx = math.inf, -1
        if local_idx >= 1 << 53:
            raise ValueError("flat index does not fit a double exactly")
        pair = torch.tensor([v, float(local_idx)], dtype=torch.float64, device=device)
    gathered = torch.empty((world, 2), dtype=torch.float64, device=pair.device)
    try:
        dist.all_gather_into_tensor(gathered, pair, group=group)
    except (RuntimeError, AttributeError, NotImplementedError):  # a backend without the flat variant
        parts = [torch.empty(2, dtype=torch.float64, device=pair.device) for _ in range(world)]
        dist.all_gather(parts, pair, group=group)
        gathered = torch.stack(parts)
    best, arg = _scan_pairs_tensor(gathered).cpu().tolist()  # the only copy to the host: 16 bytes
    return best, int(arg)


# --------------------------------------------------------------------------- dense grid
class DenseGrid:
    """A block of a dense parameter grid on one GPU (covest_grid* of the C ABI).

    axes: one sequence per model parameter (a fixed parameter is a 1-element axis).
    flat_range: (begin, end) of the flat itertools.product indices this object
    evaluates; None = the whole grid.
    """

    def __init__(self, model, axes, flat_range=None):
        if len(axes) != model.param_count:
            raise ValueError("need one axis per model parameter")
        self.model = model
        ptrs, lens = self._take_axes(axes, flat_range)
        L = _capi.lib()
        h = ctypes.c_void_p()
        _capi.check(L.covest_grid_create(model.handle, len(self.axes), ptrs.ctypes.data, lens.ctypes.data, self.flat_range[0],
                                         self.flat_range[1], ctypes.byref(h)), "covest_grid_create")
        self._handle = h
        if hasattr(model, "_register_grid"):
            model._register_grid(self)

    def _take_axes(self, axes, flat_range):
        """The axes as ONE contiguous float64 buffer (self.axes are views into it) and what the C ABI wants of them:
        an array of the axes' addresses and one of their lengths.  Kept cheap on purpose -- optimize_grid does this
        every iteration."""
        parts = [a if type(a) is np.ndarray and a.dtype == np.float64 and a.ndim == 1
                 else np.asarray(a, dtype=np.float64).reshape(-1) for a in axes]
        lens = [p.shape[0] for p in parts]
        flat = np.concatenate(parts) if len(parts) > 1 else np.ascontiguousarray(parts[0])
        self._flat = flat
        self.axes, offs, at = [], [], 0
        for n in lens:
            self.axes.append(flat[at:at + n])
            offs.append(at)
            at += n
        self.shape = tuple(lens)
        self.total = math.prod(lens)
        begin, end = (0, self.total) if flat_range is None else flat_range
        self.flat_range = (int(begin), int(end))
        base = flat.ctypes.data
        self._abi_axes = (np.array([base + 8 * o for o in offs], dtype=np.uint64), np.array(lens, dtype=np.int64))
        return self._abi_axes

    def reset(self, axes, flat_range=None):
        """Other axes and/or another block on the SAME handle (covest_grid_reset): the device memory stays; what
        optimize_grid does between its iterations."""
        if len(axes) != self.model.param_count:
            raise ValueError("need one axis per model parameter")
        ptrs, lens = self._take_axes(axes, flat_range)
        _capi.check(_capi.lib().covest_grid_reset(self._handle, len(self.axes), ptrs.ctypes.data, lens.ctypes.data,
                                                  self.flat_range[0], self.flat_range[1]), "covest_grid_reset")
        return self

    def close(self):
        if getattr(self, "_handle", None) is not None:
            _capi.lib().covest_grid_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.flat_range[1] - self.flat_range[0]

    def evaluate(self, kernel="auto", stream=None, scan_start=None):
        """Launch LL + arg-min for the block (asynchronous on `stream`, a raw
        hipStream_t value such as torch.cuda.current_stream().cuda_stream).  scan_start: also run the selection
        scan of covest/grid.py:65-70 from that minimum on the device (covest_grid_eval_scan); scan_records()
        returns what it found."""
        if stream:
            _capi.require_shared_runtime("DenseGrid.evaluate(stream=...)")
        if scan_start is None:
            _capi.check(_capi.lib().covest_grid_eval(self._handle, _capi.KERNELS[kernel],
                                                     ctypes.c_void_p(stream or 0)), "covest_grid_eval")
        else:
            _capi.check(_capi.lib().covest_grid_eval_scan(self._handle, _capi.KERNELS[kernel], ctypes.c_void_p(stream or 0),
                                                          float(scan_start)), "covest_grid_eval_scan")

    _SCAN_CAP = 128

    def scan_records(self):
        """(flat indices, -LL values) of the strict running-minimum records below the scan_start of the last
        evaluate(), in index order -- where the reference's selection loop changes its state --, or None when the
        device's list is incomplete (the caller reads the values back: loglikelihoods())."""
        if not hasattr(self, "_scan_buf"):
            self._scan_buf = ((ctypes.c_int64 * self._SCAN_CAP)(), (ctypes.c_double * self._SCAN_CAP)(),
                              ctypes.c_int32(), ctypes.c_int32())
        idx, val, n, trunc = self._scan_buf
        _capi.check(_capi.lib().covest_grid_scan(self._handle, self._SCAN_CAP, idx, val, ctypes.byref(n),
                                                 ctypes.byref(trunc)), "covest_grid_scan")
        if trunc.value:
            return None
        return idx[:n.value], val[:n.value]

    def argmin(self):
        """(min -LL, global flat index) of the last evaluate(); index -1 if none < +inf."""
        v = ctypes.c_double()
        i = ctypes.c_int64()
        _capi.check(_capi.lib().covest_grid_argmin(self._handle, ctypes.byref(v), ctypes.byref(i)),
                    "covest_grid_argmin")
        return v.value, i.value

    def loglikelihoods(self):
        """LL of every point of the block (host ndarray, flat order)."""
        out = np.empty(len(self), dtype=np.float64)
        _capi.check(_capi.lib().covest_grid_ll_host(self._handle, out.ctypes.data_as(_DP)),
                    "covest_grid_ll_host")
        return out

    @property
    def ll_device_ptr(self):
        return _capi.lib().covest_grid_ll_device(self._handle)

    def argmin_pair_tensor(self, device_index):
        """The reduction of the last evaluate() where the arg-min kernel left it: a 2-element float64 torch
        tensor {min -LL, GLOBAL flat index (-1: none)} aliasing the handle's 16 bytes in HBM (no copy, no
        synchronisation: consume it on the stream evaluate() ran on)."""
        import torch
        _capi.require_shared_runtime("DenseGrid.argmin_pair_tensor")
        ptr = _capi.lib().covest_grid_argmin_pair_device(self._handle)

        class _View:  # the CUDA array interface is how torch adopts foreign device memory (HIP included)
            __cuda_array_interface__ = {"shape": (2,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

        return torch.as_tensor(_View(), device=torch.device("cuda", device_index))

    def work(self):
        """(pmf terms, algorithmic flops, kernel name) of the last evaluate()."""
        t, f, name = ctypes.c_double(), ctypes.c_double(), ctypes.c_char_p()
        _capi.check(_capi.lib().covest_grid_work(self._handle, ctypes.byref(t), ctypes.byref(f),
                                                 ctypes.byref(name)), "covest_grid_work")
        return t.value, f.value, (name.value or b"").decode()

    def profile(self, enable=True):
        """Bracket every likelihood launch with hipEvents (see kernel_ms)."""
        _capi.check(_capi.lib().covest_grid_profile(self._handle, 1 if enable else 0),
                    "covest_grid_profile")

    def kernel_ms(self):
        """(summed device ms of the likelihood kernel, launches) since profile(True)."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _capi.check(_capi.lib().covest_grid_kernel_ms(self._handle, ctypes.byref(ms), ctypes.byref(n)),
                    "covest_grid_kernel_ms")
        return ms.value, n.value

    def point(self, flat_index):
        """Parameter tuple at a flat itertools.product index."""
        idx = np.unravel_index(int(flat_index), self.shape)
        return tuple(float(a[i]) for a, i in zip(self.axes, idx))


def repeats_cost_weights(model, axes):
    """Per-point cost (T-1) over the (q1,q2,q) sub-grid, for load-balanced partitioning."""
    q = np.array(list(itertools.product(*[np.asarray(a, dtype=np.float64) for a in axes[2:5]])))
    lo = np.array([b[0] for b in model.bounds[2:5]], dtype=np.float64)
    hi = np.array([b[1] for b in model.bounds[2:5]], dtype=np.float64)
    t = model.get_hist_threshold_values(np.clip(q, lo, hi))
    return np.maximum(t.astype(np.float64) - 1.0, 0.0)


def dense_grid_argmin(model, axes, kernel="auto", group=None, balance=True, devices=None):
    """Arg-min of -LL over a dense grid, block-partitioned over the process group -- or, with `devices`, over
    several GPUs of THIS process.

    Returns (min_negll, flat_index, params).  Every rank returns the same answer.

    devices: a list of HIP ordinals.  The reference's consumer of the grid map is ONE process that fans the points out
    to workers (covest/covest.py:86-89 -> covest/grid.py:63-64); this is that shape on GPUs: one model handle and one
    grid handle per device (the C ABI takes the device per handle, include/covest_amd.h), the flat index range cut into
    contiguous blocks balanced by sum(T - 1), every block launched asynchronously on its device's default stream --
    they run side by side --, and the N 16-byte (min, index) pairs scanned on the host with the reference's rule.  No
    process group, no collective: nothing to exchange but what the arg-min kernels stored to page-locked memory.  An
    ordinal may appear more than once (two blocks on one card: how a one-GPU box tests this).
    """
    if devices is not None:
        return _dense_grid_argmin_devices(model, axes, kernel, balance, list(devices))
    import torch.distributed as dist
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    shape = [len(a) for a in axes]
    total = int(np.prod(shape, dtype=np.int64))
    weights = None
    if world > 1 and balance and model.param_count == 5:
        weights = repeats_cost_weights(model, axes)
    bounds = partition_flat_range(total, world, weights)
    grid = DenseGrid(model, axes, (bounds[rank], bounds[rank + 1]))
    try:
        if world > 1 and dist.get_backend(group) == "nccl":
            # likelihood, arg-min, exchange and scan all on the current stream; one 16-byte copy at the end
            import torch
            grid.evaluate(kernel=kernel, stream=torch.cuda.current_stream().cuda_stream)
            gmin, gidx = distributed_argmin(None, None, group=group,
                                            pair=grid.argmin_pair_tensor(torch.cuda.current_device()))
        else:
            grid.evaluate(kernel=kernel)
            local_min, local_idx = grid.argmin()
            gmin, gidx = distributed_argmin(local_min, local_idx, group=group)
        params = grid.point(gidx) if gidx >= 0 else None
    finally:
        grid.close()
    return gmin, gidx, params


class DeviceBlocks:
    """One dense grid cut into one block per entry of `devices`, all driven by this process (dense_grid_argmin with
    `devices`).  Kept as an object so that a caller that evaluates the same grid shape repeatedly (bench.py) pays for
    the handles once: evaluate() launches every block, argmin() waits for them and scans the pairs."""

    def __init__(self, model, axes, devices, balance=True):
        if not devices:
            raise ValueError("devices: at least one HIP ordinal")
        self.devices = [int(d) for d in devices]
        shape = [len(a) for a in axes]
        total = int(np.prod(shape, dtype=np.int64))
        weights = repeats_cost_weights(model, axes) if balance and model.param_count == 5 and len(devices) > 1 else None
        self.bounds = partition_flat_range(total, len(self.devices), weights)
        self._own = []   # the per-device copies of the model this object made (closed with it)
        self.grids = []
        for i, d in enumerate(self.devices):
            m = model.on_device(d) if hasattr(model, "on_device") else model
            if m is not model:
                self._own.append(m)
            self.grids.append(DenseGrid(m, axes, (self.bounds[i], self.bounds[i + 1])))

    def evaluate(self, kernel="auto"):
        for g in self.grids:  # asynchronous: the devices work side by side
            g.evaluate(kernel=kernel)

    def argmin(self):
        """(min -LL, global flat index): every block's pair, scanned in block order with the reference's rule."""
        return scan_min_pairs([g.argmin() for g in self.grids])

    def point(self, flat_index):
        return self.grids[0].point(flat_index)

    def close(self):
        for g in self.grids:
            g.close()
        for m in self._own:
            m.close()
        self.grids, self._own = [], []


def _dense_grid_argmin_devices(model, axes, kernel, balance, devices):
    blocks = DeviceBlocks(model, axes, devices, balance=balance)
    try:
        blocks.evaluate(kernel=kernel)
        gmin, gidx = blocks.argmin()
        params = blocks.point(gidx) if gidx >= 0 else None
    finally:
        blocks.close()
    return gmin, gidx, params


# --------------------------------------------------------------------------- optimize_grid
def first_wins_scan(vals, min_val, sgn=1):
    """The selection loop of covest/grid.py:65-70 over a value array.

    Returns (min_val, argmin or -1, diff) exactly as the sequential scan would:
    only strict improvements are visited, in index order, and `diff`
    accumulates `min_val - val` at each of them in that order.
    (Round 4: only the values below the STARTING minimum can ever pass the test -- the running minimum only falls --, and
    after a search's first iterations those are a handful: they are picked out with one vectorised compare and walked
    by the reference's own loop; where they are many, the strict running-minimum records among them are picked first.
    59 -> 12 us for an iteration's 7 776 values, a fifth of an optimize_grid search.)
    """
    v = np.asarray(vals, dtype=np.float64)
    sv = v if sgn == 1 else sgn * v
    diff = 0.0
    arg = -1
    with np.errstate(invalid="ignore"):
        cand = np.flatnonzero(sv < min_val)  # (a NaN never passes: NaN < x is False)
    if cand.size > 256:
        sc = sv[cand]
        prev = np.minimum.accumulate(sc)
        cand = cand[np.concatenate(([True], sc[1:] < prev[:-1]))]
    for i in cand:
        if sv[i] < min_val:
            diff += min_val - v[i]
            min_val = sv[i]
            arg = int(i)
    return min_val, arg, diff


def replay_records(indices, vals, min_val):
    """first_wins_scan over the device's list (DenseGrid.scan_records): the loop of covest/grid.py:65-70 visits
    exactly these points with a passing test, in this order -- the same comparisons and the same sums."""
    diff = 0.0
    arg = -1
    for i, v in zip(indices, vals):
        if v < min_val:
            diff += min_val - v
            min_val = v
            arg = int(i)
    return min_val, arg, diff


def _batched_negll(fn):
    """If fn is CoverageEstimator.likelihood_f over a GPU-backed model, return a
    function evaluating a whole list of axes at once; else None."""
    est = getattr(fn, "__self__", None)
    model = getattr(est, "model", None)
    if est is None or model is None or not hasattr(model, "loglikelihood_points"):
        return None
    if getattr(fn, "__name__", "") != "likelihood_f":
        return None
    return est.negll_grid


def optimize_grid(fn, initial_guess, bounds=None, maximize=False, fix=None,
                  n_threads=constants.DEFAULT_THREAD_COUNT, trace=None, reference_specials=None):
    """covest/grid.py:17-79.  n_threads is accepted and ignored on the GPU path.  `trace`: a list the caller
    owns; one record per iteration is appended to it (what the reference prints through verbose_print,
    covest/grid.py:60,73-74) -- per call, so concurrent searches (the lock-step refinement runs threads) never
    share one.  `reference_specials` (fn = CoverageEstimator.likelihood_f only): True makes the search see what the
    REFERENCE's objective returns where its long-double pmf product overflows -- -(+inf), which the scan of
    covest/grid.py:65-70 selects, or NaN -- in THIS search: the flag travels with the evaluation requests
    (CoverageEstimator.negll_grid(..., reference_specials=)), the estimator itself is not touched, so searches and
    refinements that share it (the lock-step refinement runs threads) keep their own setting.  None: the estimator's."""
    est = getattr(fn, "__self__", None)
    def generate_axes(args, step, max_depth):
        def single(var, fixed=None):
            if fixed is None:
                return [var * step ** d for d in range(-max_depth, max_depth + 1) if d != 0]
            return [fixed]

        def within(var_grid, i):
            if bounds is None or len(bounds) <= i or len(bounds[i]) != 2:
                return var_grid
            low, high = bounds[i]
            return [var for var in var_grid
                    if (low is None or var >= low) and (high is None or var <= high)]

        return [within(single(var, fix[i]), i) for i, var in enumerate(args)]

    if fix is None:
        fix = [None] * len(initial_guess)
    sgn = -1 if maximize else 1
    batched = _batched_negll(fn)
    # the selection scan on the device (estimator.negll_grid_scan): the values stay in HBM, a handful of records come back
    scanned = getattr(est, "negll_grid_scan", None) if batched is not None and sgn == 1 else None
    if batched is not None and reference_specials is not None:
        specials = bool(reference_specials)
        batched_all, scanned_all = batched, scanned
        batched = lambda axes: batched_all(axes, reference_specials=specials)
        scanned = None if scanned_all is None else (lambda axes, mv: scanned_all(axes, mv, reference_specials=specials))
        min_val = sgn * float(est.negll_points([initial_guess], reference_specials=specials)[0])
    else:
        min_val = sgn * fn(initial_guess)
    min_args = initial_guess
    step = constants.STEP
    grid_depth = constants.GRID_DEPTH
    diff = 1
    n_iter = 0
    if trace is None:
        trace = []
    try:
        while diff > 0.1 or step > 1.001:
            n_iter += 1
            axes = generate_axes(min_args, step, grid_depth)
            n_points = int(np.prod([len(a) for a in axes], dtype=np.int64))
            records = None
            if n_points == 0:
                res = np.empty(0)
            elif scanned is not None:
                records, res = scanned(axes, min_val)
            elif batched is not None:
                res = batched(axes)
            else:
                res = np.array([fn(p) for p in itertools.product(*axes)], dtype=np.float64)
            if records is not None:
                min_val, arg, diff = replay_records(records[0], records[1], min_val)
            else:
                min_val, arg, diff = first_wins_scan(res, min_val, sgn)
            if arg >= 0:
                idx = np.unravel_index(arg, [len(a) for a in axes])
                min_args = tuple(a[i] for a, i in zip(axes, idx))
            if diff < 1.0:
                step = 1 + (step - 1) * 0.75
            trace.append({"iter": n_iter, "grid_size": n_points, "diff": diff, "step": step,
                          "args": tuple(min_args), "value": min_val})
    except KeyboardInterrupt:
        pass
    return min_args


def initial_grid(initial_guess, count=constants.INITIAL_GRID_COUNT, bounds=None, fix=None):
    """covest/grid.py:82-114: `count` random multi-start seeds around the guess
    (point generation only; no likelihood is evaluated here)."""
    if fix is None:
        fix = [None] * len(initial_guess)
    step = constants.INITIAL_GRID_STEP

    def apply_bounds(interval, i):
        if bounds is None or len(bounds) <= i or len(bounds[i]) != 2:
            return interval
        lb, rb = bounds[i]
        li, ri = interval
        if lb is not None:
            li = max(li, lb)
        if rb is not None:
            ri = min(ri, rb)
        return li, ri

    def random_params():
        intervals = [apply_bounds((var / step, var * step), i) for i, var in enumerate(initial_guess)]
        return [random.uniform(*iv) if fix[i] is None else fix[i] for i, iv in enumerate(intervals)]

    if count < 1:
        return []
    return [initial_guess] + [random_params() for _ in range(count - 1)]
