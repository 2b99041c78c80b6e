"""Build the HIP library in-tree: covest_amd/lib/libcovest_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container;
the built .so is git-ignored but travels to the GPU box with the repo snapshot.

    python -m covest_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcovest_amd.so")
# (source, extra flags, object name).  K-factored is compiled once per template variant, each into a translation unit
# of its own (csrc/ll_factored.hip, COVEST_FACTORED_VARIANT): the HIP runtime loads a translation unit's code object
# when one of its kernels is first launched, so a process only pays for the variants it uses.
SOURCES = [("host_common.cpp", (), "host_common"), ("tiles_host.cpp", (), "tiles_host"), ("plan_factored.cpp", (), "plan_factored"),
           ("abi_model.cpp", (), "abi_model"), ("abi_grid.cpp", (), "abi_grid"), ("kmer_host.cpp", (), "kmer_host"),
           ("thin_host.cpp", (), "thin_host"), ("reads_io.cpp", (), "reads_io"), ("ll_direct.hip", (), "ll_direct"),
           ("ll_basic.hip", (), "ll_basic"), ("ll_factored.hip", (), "ll_factored"), ("argmin.hip", (), "argmin"),
           ("kmer_count.hip", (), "kmer_count"), ("kmer_wide.hip", (), "kmer_wide"), ("kmer_bulk.hip", (), "kmer_bulk"), ("thin_hist.hip", (), "thin_hist")]
SOURCES += [("ll_factored.hip", ("-DCOVEST_FACTORED_VARIANT=%d" % v,), "ll_factored_v%d" % v) for v in range(10)]
SOURCES += [("ll_basic.hip", ("-DCOVEST_BASIC_VARIANT=%d" % v,), "ll_basic_v%d" % v) for v in range(8)]
MAX_PARALLEL = 8
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "covest_amd.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source for gfx950 and link the shared library.  `extra_flags` / `out`: an
    experimental variant (e.g. -DCOVEST_EXP_...) linked somewhere else, for A/B runs via COVEST_AMD_LIB."""
    if out is None and not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj" if out is None else "obj_" + os.path.basename(out))
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    common = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
              "-x", "hip"] + list(extra_flags)
    objs = []
    pending = list(SOURCES)
    running = []

    def reap(block_until_below):
        while len(running) >= block_until_below:
            src, p = running.pop(0)
            log, _ = p.communicate()
            if p.returncode != 0:
                for _, q in running:
                    q.kill()
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, log.decode(errors="replace")))
            if verbose and log:
                print(log.decode(errors="replace"))

    for src, flags, name in pending:
        obj = os.path.join(obj_dir, name + ".o")
        cmd = [hipcc] + common + list(flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        reap(MAX_PARALLEL)
        running.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    reap(1)
    target = LIB_PATH if out is None else out
    link = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", target] + objs
    subprocess.check_call(link)
    return target


if __name__ == "__main__":
    # python -m covest_amd.build [--force] [--out path.so -DFLAG ...]
    args = sys.argv[1:]
    out = args[args.index("--out") + 1] if "--out" in args else None
    flags = [a for a in args if a.startswith("-D") or a.startswith("-m")]
    print(build(force="--force" in args, verbose=out is None, extra_flags=flags, out=out))
