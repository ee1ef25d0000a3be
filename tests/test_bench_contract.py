"""The bench line's contract, checked on CPU against the newest committed line (profiles/r*_bench.json, written by
`python bench.py` on an MI355X box) and against bench.py's own argument handling: the keys the driver reads, the tier's
`roofline` / `cpu_baseline` objects, internal consistency of the numbers (value = steps * batch / time, frac = achieved /
peak), and that nothing in bench.py reads /root/reference or imports the oracle outside the cpu_baseline leg."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return files[-1], json.loads(open(files[-1]).read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_keys():
    f, d = _newest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (f, k)
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["vs_baseline"] is None and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["global_batch"]
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-3
    assert 0 < r["step_frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_bench_source_keeps_the_oracle_in_the_baseline_leg_and_off_the_reference():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "/root/reference" not in src
    # every import of the oracle sits inside the cpu_baseline helpers
    for m in re.finditer(r"^(\s*)(from oracle|import oracle)", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        assert fn in ("_oracle_step_fn", "cpu_baseline"), fn
    pkg = os.path.join(ROOT, "multimodal_vae_comparison_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import) oracle", s, re.M), os.path.join(dirpath, f)
                assert "/root/reference" not in s, os.path.join(dirpath, f)
