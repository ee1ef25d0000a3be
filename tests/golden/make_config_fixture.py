"""Dump the reference's SHIPPED run configurations as data (build container only).

    python tests/golden/make_config_fixture.py

Parses every /root/reference/multimodal_compare/configs/config_*.yml and the `feature_dims` table of the dataset
class each one names (models/datasets.py, looked up as models/dataloader.py:40-41 does) and writes
tests/golden/shipped_configs.json: {file name: {"config": parsed YAML, "feature_dims": {...}}}.  The CPU test
tests/test_shipped_configs.py re-creates each YAML file from this and constructs MultimodalVAE from it, so that
"configs/config_*.yml still select the path unchanged" (BASELINE north_star) is checked wherever the tests run.
"""
import glob
import json
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install()
from models import datasets  # noqa: E402  (the reference package)

out = {}
for path in sorted(glob.glob(os.path.join(ref_harness.REF_ROOT, "configs", "config_*.yml"))):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cls = getattr(datasets, cfg["dataset_name"].upper())
    out[os.path.basename(path)] = {"config": cfg, "feature_dims": {k: list(v) for k, v in cls.feature_dims.items()}}
with open(os.path.join(HERE, "shipped_configs.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print({k: (v["config"]["mixing"], v["config"]["obj"]) for k, v in out.items()})
