"""ctypes binding of include/covest_amd.h (covest_amd/lib/libcovest_amd.so).

Loading the library initialises nothing on the GPU (fork-safe, like importing the
reference's `covest_poisson` extension); the first compute call does.  There is
no CPU fallback: if the library is missing or no HIP device is usable the call
raises CovestHipError -- it never silently computes elsewhere.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COVEST_AMD_LIB", os.path.join(_HERE, "lib", "libcovest_amd.so"))

MAX_PARAMS = 5
MODEL_BASIC, MODEL_REPEATS = 0, 1
KERNEL_AUTO, KERNEL_DIRECT, KERNEL_RECUR, KERNEL_FACTORED, KERNEL_DIRECT_REF = 0, 1, 2, 3, 4
KERNELS = {"auto": KERNEL_AUTO, "direct": KERNEL_DIRECT, "recur": KERNEL_RECUR,
           "factored": KERNEL_FACTORED, "direct_ref": KERNEL_DIRECT_REF}

# every symbol include/covest_amd.h declares (tests check the library exports them all)
EXPORTS = (
    "covest_abi_version", "covest_device_count", "covest_last_error",
    "covest_model_create", "covest_model_destroy", "covest_model_param_count",
    "covest_model_bins_evaluated", "covest_threshold_o", "covest_eval_points",
    "covest_probabilities", "covest_reference_overflow", "covest_grid_create", "covest_grid_reset", "covest_grid_destroy", "covest_grid_size",
    "covest_grid_eval", "covest_grid_eval_scan", "covest_grid_scan", "covest_grid_argmin", "covest_grid_argmin_pair_device",
    "covest_grid_ll_device",
    "covest_grid_ll_host",
    "covest_grid_work", "covest_grid_profile", "covest_grid_kernel_ms", "covest_grid_diag",
    "covest_kmer_create", "covest_kmer_destroy", "covest_kmer_reserve", "covest_kmer_add",
    "covest_kmer_add_device", "covest_kmer_histogram", "covest_kmer_slots", "covest_kmer_clear",
    "covest_kmer_count_reads_device", "covest_kmer_partition_info", "covest_kmer_partition_ms", "covest_kmer_memory_limit",
    "covest_kmer_scatter_rate",
    "covest_reads_open", "covest_reads_close", "covest_reads_next", "covest_reads_bytes",
    "covest_thin_histogram", "covest_thin_histogram_timed",
)


class CovestHipError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("k", ctypes.c_int32),
        ("r", ctypes.c_int32),
        ("n_err", ctypes.c_int32),
        ("comb", ctypes.POINTER(ctypes.c_double)),
        ("n_keys", ctypes.c_int64),
        ("keys", ctypes.POINTER(ctypes.c_int32)),
        ("counts", ctypes.POINTER(ctypes.c_double)),
        ("tail", ctypes.c_double),
        ("lo", ctypes.c_double * MAX_PARAMS),
        ("hi", ctypes.c_double * MAX_PARAMS),
        ("threshold", ctypes.c_double),
        ("has_threshold", ctypes.c_int32),
        ("device", ctypes.c_int32),
    ]


_lib = None
# One process hosts ONE HIP runtime only if torch (which ships its own copy of libamdhip64) was imported BEFORE this
# library was loaded: then the loader binds this library to the runtime torch brought.  The other way round the
# process ends up with two runtimes, and device pointers of one mean nothing to the other.
_loaded_before_torch = False


def lib():
    """The loaded library; raises CovestHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CovestHipError(
            "HIP library %s is missing: run `python -m covest_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    global _loaded_before_torch
    _loaded_before_torch = "torch" not in sys.modules
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise CovestHipError("cannot load %s: %s" % (LIB_PATH, e))
    if "COVEST_AMD_LIB" in os.environ:
        # an A/B run against an OLDER build (tools/ab.sh): an entry point that build does not have yet is bound to a
        # stub that fails loudly when called -- the shipped library (no override) must export every one of them
        class _Tolerant:
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    def missing(*a, **k):
                        raise CovestHipError("%s is not exported by %s" % (name, LIB_PATH))
                    missing.restype = missing.argtypes = None
                    object.__setattr__(self, name, missing)
                    return missing

        L = _Tolerant(L)
    vp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)
    i32, i64 = ctypes.c_int32, ctypes.c_int64
    L.covest_abi_version.restype = ctypes.c_int
    L.covest_abi_version.argtypes = []
    L.covest_device_count.restype = ctypes.c_int
    L.covest_device_count.argtypes = []
    L.covest_last_error.restype = ctypes.c_char_p
    L.covest_last_error.argtypes = []
    L.covest_model_create.restype = ctypes.c_int
    L.covest_model_create.argtypes = [ctypes.POINTER(ModelDesc), ctypes.POINTER(vp)]
    L.covest_model_destroy.restype = None
    L.covest_model_destroy.argtypes = [vp]
    L.covest_model_param_count.restype = ctypes.c_int
    L.covest_model_param_count.argtypes = [vp]
    L.covest_model_bins_evaluated.restype = i64
    L.covest_model_bins_evaluated.argtypes = [vp]
    L.covest_threshold_o.restype = ctypes.c_int
    L.covest_threshold_o.argtypes = [i64, dp, ctypes.c_double, i32, i32, ctypes.POINTER(i32)]
    L.covest_eval_points.restype = ctypes.c_int
    L.covest_eval_points.argtypes = [vp, i64, dp, dp, i32]
    L.covest_reference_overflow.restype = ctypes.c_int
    L.covest_reference_overflow.argtypes = [vp, i64, dp, ctypes.POINTER(ctypes.c_uint8)]
    L.covest_probabilities.restype = ctypes.c_int
    L.covest_probabilities.argtypes = [vp, dp, i32, dp]
    L.covest_grid_create.restype = ctypes.c_int
    # (the axes' pointer array and the lengths as plain addresses: DenseGrid hands over two numpy arrays, one
    # conversion each, instead of a ctypes object per axis -- an optimize_grid iteration re-configures its handle)
    L.covest_grid_create.argtypes = [vp, i32, vp, vp, i64, i64, ctypes.POINTER(vp)]
    L.covest_grid_reset.restype = ctypes.c_int
    L.covest_grid_reset.argtypes = [vp, i32, vp, vp, i64, i64]
    L.covest_grid_destroy.restype = None
    L.covest_grid_destroy.argtypes = [vp]
    L.covest_grid_size.restype = i64
    L.covest_grid_size.argtypes = [vp]
    L.covest_grid_eval.restype = ctypes.c_int
    L.covest_grid_eval.argtypes = [vp, i32, vp]
    L.covest_grid_eval_scan.restype = ctypes.c_int
    L.covest_grid_eval_scan.argtypes = [vp, i32, vp, ctypes.c_double]
    L.covest_grid_scan.restype = ctypes.c_int
    L.covest_grid_scan.argtypes = [vp, i32, ctypes.POINTER(i64), dp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.covest_grid_argmin.restype = ctypes.c_int
    L.covest_grid_argmin.argtypes = [vp, dp, ctypes.POINTER(i64)]
    L.covest_grid_argmin_pair_device.restype = vp
    L.covest_grid_argmin_pair_device.argtypes = [vp]
    L.covest_grid_ll_device.restype = vp
    L.covest_grid_ll_device.argtypes = [vp]
    L.covest_grid_ll_host.restype = ctypes.c_int
    L.covest_grid_ll_host.argtypes = [vp, dp]
    L.covest_grid_work.restype = ctypes.c_int
    L.covest_grid_work.argtypes = [vp, dp, dp, ctypes.POINTER(ctypes.c_char_p)]
    L.covest_grid_profile.restype = ctypes.c_int
    L.covest_grid_profile.argtypes = [vp, i32]
    L.covest_grid_kernel_ms.restype = ctypes.c_int
    L.covest_grid_kernel_ms.argtypes = [vp, dp, ctypes.POINTER(i64)]
    u8p = ctypes.POINTER(ctypes.c_uint8)
    i64p = ctypes.POINTER(i64)
    L.covest_kmer_create.restype = ctypes.c_int
    L.covest_kmer_create.argtypes = [i32, i32, i64, i32, ctypes.POINTER(vp)]
    L.covest_kmer_destroy.restype = None
    L.covest_kmer_destroy.argtypes = [vp]
    L.covest_kmer_reserve.restype = ctypes.c_int
    L.covest_kmer_reserve.argtypes = [vp, i64]
    L.covest_kmer_add.restype = ctypes.c_int
    L.covest_kmer_add.argtypes = [vp, u8p, i64p, i64]
    L.covest_kmer_add_device.restype = ctypes.c_int
    L.covest_kmer_add_device.argtypes = [vp, vp, vp, i64, i64, vp]
    L.covest_kmer_count_reads_device.restype = ctypes.c_int
    L.covest_kmer_count_reads_device.argtypes = [vp, vp, vp, i64, i64, i64, vp]
    L.covest_kmer_partition_info.restype = ctypes.c_int
    L.covest_kmer_partition_info.argtypes = [vp, i64p]
    L.covest_kmer_memory_limit.restype = ctypes.c_int
    L.covest_kmer_memory_limit.argtypes = [vp, i64]
    L.covest_kmer_partition_ms.restype = ctypes.c_int
    L.covest_kmer_partition_ms.argtypes = [vp, dp]
    L.covest_kmer_scatter_rate.restype = ctypes.c_int
    L.covest_kmer_scatter_rate.argtypes = [i32, i64, i64, dp]
    L.covest_kmer_histogram.restype = ctypes.c_int
    L.covest_kmer_histogram.argtypes = [vp, i64p, i64, i64p, i64p]
    L.covest_kmer_clear.restype = ctypes.c_int
    L.covest_kmer_clear.argtypes = [vp, vp]
    L.covest_kmer_slots.restype = i64
    L.covest_kmer_slots.argtypes = [vp]
    L.covest_reads_open.restype = ctypes.c_int
    L.covest_reads_open.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_uint64, ctypes.POINTER(vp)]
    L.covest_reads_close.restype = None
    L.covest_reads_close.argtypes = [vp]
    L.covest_reads_next.restype = ctypes.c_int
    L.covest_reads_next.argtypes = [vp, i64, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(i64p), i64p]
    L.covest_reads_bytes.restype = i64
    L.covest_reads_bytes.argtypes = [vp]
    L.covest_thin_histogram.restype = ctypes.c_int
    L.covest_thin_histogram.argtypes = [ctypes.c_int32, i64, ctypes.POINTER(ctypes.c_int32), dp, ctypes.c_double, i64, dp]
    L.covest_thin_histogram_timed.restype = ctypes.c_int
    L.covest_thin_histogram_timed.argtypes = [ctypes.c_int32, i64, ctypes.POINTER(ctypes.c_int32), dp, ctypes.c_double,
                                              i64, dp, ctypes.c_int32, dp]
    L.covest_grid_diag.restype = i64
    L.covest_grid_diag.argtypes = [vp, ctypes.POINTER(i64), i64]
    _lib = L
    return L


COVEST_E_INVALID = -1
COVEST_E_NOMEM = -4
COVEST_E_UNSUPPORTED = -5


def hip_runtimes_mapped():
    """The distinct libamdhip64 files mapped into this process (/proc/self/maps; inode-distinct paths): one when torch
    and this library bind the same runtime, two when the process hosts torch's bundled copy beside the system's."""
    seen = {}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 6 and "libamdhip64" in os.path.basename(parts[5]):
                    seen[(parts[3], parts[4])] = parts[5]  # (device, inode): one entry per file
    except OSError:
        return None
    return sorted(seen.values())


def require_shared_runtime(what):
    """Called wherever device memory of this library is handed to torch (or torch's to it): fails loudly if the process
    hosts TWO HIP runtimes -- the pointers of one mean nothing to the other.  Decided from what is mapped
    (hip_runtimes_mapped: two different libamdhip64 files); where /proc is not readable, from the import order, which
    is how a process ends up with two (torch ships its own copy and must be imported BEFORE this library is loaded,
    INTEGRATION.md) -- a torch built against the system's ROCm binds the same file whatever the order, and is fine."""
    if _lib is None or "torch" not in sys.modules:
        return
    mapped = hip_runtimes_mapped()
    two = len(mapped) > 1 if mapped is not None else _loaded_before_torch
    if two:
        raise CovestHipError(
            "%s: this process hosts two HIP runtimes (%s), so device pointers cannot be shared between torch and "
            "libcovest_amd.so -- `import torch` before the first covest_amd call%s (INTEGRATION.md)"
            % (what, ", ".join(mapped) if mapped else "torch was imported after the library was loaded",
               "" if not _loaded_before_torch else ": here torch was imported after the library was loaded"))


def last_error():
    return (lib().covest_last_error() or b"").decode()


def check(status, what):
    if status != 0:
        msg = lib().covest_last_error()
        raise CovestHipError("%s failed (%d): %s" % (what, status, (msg or b"").decode()))


def device_count():
    n = lib().covest_device_count()
    if n < 0:
        raise CovestHipError("no usable HIP device: %s" % lib().covest_last_error().decode())
    return n
