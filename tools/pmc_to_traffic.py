#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of a round (tools/pmc_profile.sh + tools/pmc_summary.py):
per workload and kernel the HBM bytes per launch, (2 FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 -- and the fp64 flops the kernel EXECUTED per launch,
64 lanes x (2 FMA + MUL + ADD wave instructions) + 2048 per v_mfma_f64_16x16x4 (SQ_INSTS_VALU_MFMA_MOPS_F64 / 4).
Every workload's entry carries the hash of the sources its counters were taken on -- config.library_source_sha16 of the
bench line the profiled run itself printed (profiles/<tag>_bench_<workload>_under_rocprof.json) --: bench.py marks its
executed-flop fraction stale when the library it measures was built from other sources.
    python tools/pmc_to_traffic.py r05"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import source_sha16  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
# workload -> (short kernel name of covest_grid_work, prefix of the profiled kernel's name)
want = {"c3": ("ll_factored", "ll_factored_kernel<512,3,false,true"), "c2": ("ll_basic", "ll_basic_kernel<false>"),
        "c3t": ("ll_factored", "ll_factored_kernel<512,3,true,true"), "c2t": ("ll_basic", "ll_basic_kernel<true>")}
out = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 and executed fp64 flops per launch from separate "
                "rocprofv3 --pmc passes (tools/pmc_profile.sh, profiles/%s_<workload>_pmc_summary.json; tools/pmc_to_traffic.py); "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced read; uncalibrated for these "
                "kernels' small reads, so the read side is an upper bound); the write side above the 8 B/point output is the "
                "scratch of spilled registers" % tag,
       }
for w, (short, full) in want.items():
    path = os.path.join(REPO, "profiles", "%s_%s_pmc_summary.json" % (tag, w))
    if not os.path.exists(path):
        continue
    with open(path) as f:
        summary = json.load(f)
    d = next((v for k, v in summary.items() if k.startswith(full)), None)
    if d is None:
        continue
    flops = 64.0 * (2.0 * d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_ADD_F64"]) \
        + 2048.0 * d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / 4.0
    sha = None
    try:  # the sources of the library that was profiled, as the profiled run reported them
        with open(os.path.join(REPO, "profiles", "%s_bench_%s_under_rocprof.json" % (tag, w))) as f:
            sha = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])["config"].get("library_source_sha16")
    except (OSError, ValueError, IndexError, KeyError):
        pass
    out[w] = {"_source_sha16": sha or source_sha16(),
              short: int(round((2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0)),
              short + "_executed_flops": flops,
              short + "_valu_instructions": d["SQ_INSTS_VALU"], short + "_mfma_instructions": d.get("SQ_INSTS_MFMA", 0.0)}
with open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
