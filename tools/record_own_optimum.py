#!/usr/bin/env python3
"""GPU box: the end point of THIS library's default flow (tests/flow_helper.py estimate: guess, L-BFGS-B refinement) on
the reference's own test histogram, for both models -> gpurun_out/own_optimum.json.  NOT reference values: where
L-BFGS-B stops on the flat ridge hangs on the last bits of the likelihood values, so the reference's end
point is not reproducible; this file pins the library's OWN end point as a regression value
(tests/golden/own_optimum.json, tests/test_gpu_hist_steps.py::test_whole_default_flow) -- a kernel change that moves
the last bits may move it, and must then re-record it knowingly."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    from flow_helper import estimate
    path = os.path.join(REPO, "tests", "golden", "sim_c10_e0.05.hist")
    out = {"what": "tests/flow_helper.py estimate on sim_c10_e0.05.hist: this library's own end point (regression value, "
                   "not the reference's)", "models": {}}
    for model in ("basic", "repeats"):
        runs = [estimate(path, model=model) for _ in range(2)]
        keys = ("coverage", "error_rate", "q1", "q2", "q", "loglikelihood", "genome_size")
        assert all(runs[0].get(k) == runs[1].get(k) for k in keys), "the flow is not deterministic"
        out["models"][model] = {k: runs[0][k] for k in keys if k in runs[0]}
        print(model, out["models"][model])
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "own_optimum.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
