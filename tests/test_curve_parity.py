"""ELBO-curve parity, shortened (tools/curve_parity.py runs the 200-step version and writes
profiles/r02_curve_parity.json): train-mode steps of the MoPoE CdSprites+ model on the HIP path (captured step,
device noise + dropout generators) and on the CPU oracle (torch generators), same initial parameters and batch
sequence per seed index.  The HIP seed-mean of the moving-average ELBO must lie inside the oracle's seed envelope
(reference loop: models/trainer.py:117-128; SURVEY 8(d) "Parity procedure")."""
import importlib.util
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _tool():
    spec = importlib.util.spec_from_file_location("curve_parity", os.path.join(ROOT, "tools", "curve_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_elbo_curves_overlap_the_oracle_envelope(hip_lib):
    cp = _tool()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    steps, seeds, B, T, D, lr = 40, 3, 64, 16, 16, 1e-3
    data = cp.make_data(4, B, T)
    hip = [cp.run_hip(s, data, steps, lr, D) for s in range(seeds)]
    orc_ = [cp.run_oracle(s, data, steps, lr, D) for s in range(seeds)]
    res = cp.compare(hip, orc_, window=8, skip=4)
    print({k: (round(v["fraction_inside"], 3), round(v["worst_relative_excursion"], 5)) for k, v in res.items()})
    for name in ("loss", "recon_mod_1", "recon_mod_2"):
        assert res[name]["fraction_inside"] >= 0.9, (name, res[name])
        assert res[name]["worst_relative_excursion"] < 0.02, (name, res[name])
    # both paths actually train
    assert res["loss"]["hip_mean_last"] < res["loss"]["hip_mean_first"]
    assert res["loss"]["oracle_mean_last"] < res["loss"]["oracle_mean_first"]
    # step 0 is deterministic up to noise: the very first losses agree to the noise level
    first_h = sum(r[0][0] for r in hip) / seeds
    first_o = sum(r[0][0] for r in orc_) / seeds
    assert abs(first_h - first_o) / abs(first_o) < 0.01
