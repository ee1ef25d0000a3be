"""gpurun_out/relaxed_bars.jsonl (written by tests/test_parity_e2e.py::_relaxed during a -m gpu run) ->
profiles/r06_relaxed_bars.json: every comparison of the GPU suite that did not pass on the plain 1e-4 bar against the
plain oracle -- which test case, which tensor, its error, the bar it was held to and why."""
import collections
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "relaxed_bars.jsonl")
allrows = [json.loads(l) for l in open(src)] if os.path.exists(src) else []
kinks = [r for r in allrows if r.get("kind") == "relu-kink-flips"]     # _assert_relu_masks_near_kinks: one per e2e case
rows = [r for r in allrows if r.get("kind") != "relu-kink-flips"]
per_case = collections.Counter(r["case"] for r in rows if r["bar"] > 1e-4 and r.get("kind") != "whole-gradient")
out = {"what": "every comparison of `pytest -m gpu` that passed on anything but the plain 1e-4 bar against the plain oracle",
       "caps": {"max_bar": 5e-3, "max_loosened_tensors_per_case": 3},
       "n_records": len(rows), "loosened_tensors_per_case": dict(per_case),
       "max_bar_used": max([r["bar"] for r in rows if r.get("kind") != "whole-gradient"], default=None),
       "relu_kink_checks": {
           "what": "per e2e case: ReLU decisions of the HIP path and of the fp32 oracle against `pre64 > 0` of the fp64 "
                   "oracle; bounds: <= 8 differing elements per site, each with |pre64| <= 1e-5 of the layer maximum",
           "n_cases": len(kinks), "elements_checked": sum(r["elements_checked"] for r in kinks),
           "cases_with_flips": {r["case"]: r["flips"] for r in kinks if r["flips"]}},
       "records": rows}
dst = os.path.join(ROOT, "gpurun_out", "r06_relaxed_bars.json")
json.dump(out, open(dst, "w"), indent=1)
print(f"{len(rows)} relaxed comparisons; loosened per case: {dict(per_case)}; max bar {out['max_bar_used']}")
