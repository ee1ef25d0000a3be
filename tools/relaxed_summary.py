"""gpurun_out/relaxed_bars.jsonl (written by tests/test_parity_e2e.py::_relaxed during a -m gpu run) ->
profiles/r05_relaxed_bars.json: every comparison of the GPU suite that did not pass on the plain 1e-4 bar against the
plain oracle -- which test case, which tensor, its error, the bar it was held to and why."""
import collections
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "relaxed_bars.jsonl")
rows = [json.loads(l) for l in open(src)] if os.path.exists(src) else []
per_case = collections.Counter(r["case"] for r in rows if r["bar"] > 1e-4 and r.get("kind") != "whole-gradient")
out = {"what": "every comparison of `pytest -m gpu` that passed on anything but the plain 1e-4 bar against the plain oracle",
       "caps": {"max_bar": 5e-3, "max_loosened_tensors_per_case": 3},
       "n_records": len(rows), "loosened_tensors_per_case": dict(per_case),
       "max_bar_used": max([r["bar"] for r in rows if r.get("kind") != "whole-gradient"], default=None), "records": rows}
dst = os.path.join(ROOT, "gpurun_out", "r05_relaxed_bars.json")
json.dump(out, open(dst, "w"), indent=1)
print(f"{len(rows)} relaxed comparisons; loosened per case: {dict(per_case)}; max bar {out['max_bar_used']}")
