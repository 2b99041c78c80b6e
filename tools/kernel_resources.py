#!/usr/bin/env python3
"""Register / spill / LDS / occupancy table of every gfx950 kernel (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py [file.hip ...]      (default: every .hip under covest_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(REPO, "covest_amd", "csrc", "*.hip")))
print("%-60s %5s %5s %6s %6s %7s %4s %8s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
# K-factored and K-basic are compiled once per template variant (covest_amd/build.py): the kernels live in those
# translation units, one -D each
VARIANTS = {"ll_factored.hip": ["-DCOVEST_FACTORED_VARIANT=%d" % v for v in range(10)],
            "ll_basic.hip": ["-DCOVEST_BASIC_VARIANT=%d" % v for v in range(8)]}
jobs = []
for f in files:
    jobs.append((f, []))
    for flag in VARIANTS.get(os.path.basename(f), []):
        jobs.append((f, [flag]))
seen = set()
for f, flags in jobs:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip"] + flags +
                         ["-c", f, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: +([^:]+(?:\[[^\]]*\])?): *(.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k.startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = name.replace("covest::(anonymous namespace)::", "").replace("void ", "")
            name = re.sub(r"\(covest::.*", "", name)
            if name in seen:  # (the finishing kernels of the common translation unit show up once)
                continue
            seen.add(name)
            print("%-60s %5s %5s %6s %6s %7s %4s %8s" % (name[:60], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("VGPRs Spill"),
                                                        cur.get("SGPRs Spill"), cur.get("ScratchSize [bytes/lane]"),
                                                        cur.get("Occupancy [waves/SIMD]"), v))
