#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of a round (tools/pmc_profile.sh + tools/pmc_summary.py):
per workload and kernel the HBM bytes per launch, (2 FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 -- and the fp64 flops the kernel EXECUTED per launch,
64 lanes x (2 FMA + MUL + ADD wave instructions) + 2048 per v_mfma_f64_16x16x4 (SQ_INSTS_VALU_MFMA_MOPS_F64 / 4).
    python tools/pmc_to_traffic.py r03"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
want = {"c3": ("ll_factored", "ll_factored_kernel<512,3,false,true>"), "c2": ("ll_basic", "ll_basic_kernel<false>")}
out = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 and executed fp64 flops per launch from separate "
                "rocprofv3 --pmc passes (tools/pmc_profile.sh, profiles/%s_c{3,2}_pmc_summary.json; tools/pmc_to_traffic.py); "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced read; uncalibrated for these "
                "kernels' small reads, so the read side is an upper bound); the write side above the 8 B/point output is the "
                "scratch of spilled registers" % tag}
for w, (short, full) in want.items():
    with open(os.path.join(REPO, "profiles", "%s_%s_pmc_summary.json" % (tag, w))) as f:
        summary = json.load(f)
    # (the headline shape of K-factored is compiled with its row stride as a fifth template argument since round 4)
    d = summary.get(full) or next(v for k, v in summary.items() if k.startswith(full[:-1] + ","))
    flops = 64.0 * (2.0 * d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_ADD_F64"]) \
        + 2048.0 * d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / 4.0
    out[w] = {short: int(round((2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0)),
              short + "_executed_flops": flops,
              short + "_valu_instructions": d["SQ_INSTS_VALU"], short + "_mfma_instructions": d.get("SQ_INSTS_MFMA", 0.0)}
with open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
