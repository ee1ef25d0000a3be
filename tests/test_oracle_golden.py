"""The oracle (oracle/mmvae_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances: fp32, relative 2e-5 on scalars (the reference's
own reduction order differs), 1e-4 relative-to-norm on gradient summaries."""
import numpy as np
import pytest
import torch

from conftest import forward_cases, full_grad_cases, golden_cases, load_full_grads, load_golden
from oracle import golden_weights as gw
from oracle import mmvae_oracle as orc


def _batch(g, unimodal=False):
    if unimodal:         # exactly one of mnist / img / onehot
        if "mnist" in g:
            return {"mod_1": {"data": torch.from_numpy(g["mnist"]), "masks": None, "categorical": False}}
        if "img" in g:
            return {"mod_1": {"data": torch.from_numpy(g["img"]), "masks": None, "categorical": False}}
        return {"mod_1": {"data": torch.from_numpy(g["onehot"]), "masks": torch.from_numpy(g["mask"]), "categorical": True}}
    if "mnist" in g:     # MNIST / SVHN image pair
        return {"mod_1": {"data": torch.from_numpy(g["mnist"]), "masks": None, "categorical": False},
                "mod_2": {"data": torch.from_numpy(g["svhn"]), "masks": None, "categorical": False}}
    b = {"mod_1": {"data": torch.from_numpy(g["img"]), "masks": None, "categorical": False},
         "mod_2": {"data": torch.from_numpy(g["onehot"]), "masks": torch.from_numpy(g["mask"]), "categorical": True}}
    if "act" in g:       # third modality: action sequences (B, T, joints, feats)
        b["mod_3"] = {"data": torch.from_numpy(g["act"]), "masks": torch.from_numpy(g["amask"]), "categorical": False}
    return b


def _run(meta, g):
    shapes = (orc.vae_param_shapes(meta["mods"][0], meta["D"]) if meta["mixing"] == "vae"
              else orc.model_param_shapes(meta["mods"], meta["D"]))
    p = gw.make_params(shapes, meta["seed"], requires_grad=True)
    eps = [torch.from_numpy(g[f"eps_{i}"]) for i in range(meta["n_eps"])]
    kw = {"order": meta["order"]} if meta["mixing"] == "poe" else {}
    if "obj" in meta:       # MoE dreg / K > 1 / prior laplace / llik_scaling auto (configs/config_mnistsvhn.yml)
        kw.update(obj=meta["obj"], K=meta["K"], prior=meta["prior"])
        assert orc.resolve_llik_scaling(meta["mods"]) == pytest.approx(meta["llik"], rel=1e-12)
    out = orc.OBJECTIVES[meta["mixing"]](p, meta["mods"], _batch(g, meta["mixing"] == "vae"), eps, meta["D"],
                                         beta=meta["beta"], **kw)
    return p, out


def _close(a, b, rtol, what, floor=1e-30):
    """max|a-b| <= rtol * max(max|b|, floor).  `floor` absorbs analytically-zero gradients (e.g. a bias that
    shifts all logits of a time-softmax equally) whose computed value is pure rounding noise (~1e-7)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), floor)
    assert err <= rtol, f"{what}: rel err {err:.3e} > {rtol}"


@pytest.mark.parametrize("name", golden_cases())
def test_objective_matches_reference(name):
    meta, g = load_golden(name)
    p, out = _run(meta, g)
    _close(out["loss"].item(), g["loss"], 2e-5, "loss")
    if meta["mixing"] == "vae":      # the fixture keeps the per-sample sums of the (B, D) / (B, F) element tensors
        _close(out["kld"].detach().sum(-1).numpy(), g["kld"], 2e-5, "kld")
        _close(out["reconstruction_loss"].detach().double().sum(-1).numpy(), g["rec_0"], 2e-5, "rec_0")
        return
    _close(out["kld"].detach().numpy(), g["kld"], 2e-5, "kld")
    for i, r in enumerate(out["reconstruction_loss"]):
        _close(r.detach().numpy(), g[f"rec_{i}"], 2e-5, f"rec_{i}")
    if meta["mixing"] == "mopoe":
        fw = out["_fw"]
        for i in range(len(meta["mods"])):
            _close(fw["enc"][i][0].detach(), g[f"enc_mu_{i}"], 1e-5, f"enc_mu_{i}")
            _close(fw["enc"][i][1].detach(), g[f"enc_lv_{i}"], 1e-5, f"enc_lv_{i}")
            _close(fw["z"][i].detach(), g[f"z_{i}"], 1e-5, f"z_{i}")
            _close(gw.summarize(fw["recon"][i], 256), g[f"recon_{i}"], 1e-5, f"recon_{i}")
        _close(fw["joint"][0].detach(), g["joint_mu"], 1e-5, "joint_mu")
        _close(fw["joint"][1].detach(), g["joint_var"], 1e-5, "joint_var")
        for S, (mu, var) in fw["subsets"].items():
            key = "_".join(f"mod_{i + 1}" for i in S)
            _close(mu.detach(), g[f"subset_mu/{key}"], 1e-5, f"subset_mu/{key}")
            _close(var.detach(), g[f"subset_var/{key}"], 1e-5, f"subset_var/{key}")


@pytest.mark.parametrize("name", golden_cases())
def test_gradients_and_adam_match_reference(name):
    meta, g = load_golden(name)
    p, out = _run(meta, g)
    out["loss"].backward()
    for k, t in p.items():
        if t.grad is None:      # MoE never touches the trainable model prior (the fixture stores zeros for it)
            assert meta["mixing"] == "moe" and k == "_pz_params.1", k
            t.grad = torch.zeros_like(t)
        _close(gw.summarize(t.grad), g[f"g/{k}"], 1e-4, f"grad {k}", floor=0.02)
    state = {k: (torch.zeros_like(t), torch.zeros_like(t), torch.zeros_like(t)) for k, t in p.items()}
    orc.adam_amsgrad_step(p, {k: t.grad for k, t in p.items()}, state, meta["lr"], 1)
    # Step 1 of Adam moves every element by ~lr*sign(g): elements whose gradient is ~0 (masked keys, unused
    # embedding rows) are ill-conditioned, so pin the update on the bulk of the sampled elements.
    for k, t in p.items():
        a, b = gw.summarize(t.data)[3:], g[f"a/{k}"][3:]
        gref = np.abs(g[f"g/{k}"][3:])
        well = gref > max(1e-3 * gref.max(), 1e-5)
        bad = (np.abs(a - b) > 1e-6 * np.maximum(np.abs(b), 1e-3)) & well
        assert not bad.any(), f"adam {k}: {bad.sum()}/{well.sum()} well-conditioned elements differ"
        assert np.abs(a - b).max() <= 2.0 * meta["lr"] * 1.001, f"adam {k}: step larger than 2*lr"


@pytest.mark.parametrize("name", full_grad_cases())
def test_every_gradient_element_matches_reference(name):
    """one small case per mixer keeps the reference's FULL parameter gradients: every tensor to 1e-4 of its maximum, and
    element-wise 1e-3 on the elements within a factor 20 of that maximum"""
    meta, g = load_golden(name)
    full = load_full_grads(name)
    p, out = _run(meta, g)
    out["loss"].backward()
    assert set(full) == set(p)
    for k, t in p.items():
        a = (t.grad if t.grad is not None else torch.zeros_like(t)).double().numpy()
        b = full[k].astype(np.float64)
        assert a.shape == b.shape, k
        _close(a, b, 1e-4, f"grad {k}", floor=0.02)
        big = np.abs(b).max()
        well = np.abs(b) >= 0.05 * big
        if big > 0 and well.any():
            el = (np.abs(a - b)[well] / np.abs(b)[well]).max()
            assert el <= 1e-3, f"grad {k}: element-wise rel err {el:.2e}"


def test_chunk_bounds():
    # mixture_component_selection arithmetic (models/mmvae_models.py:396-410) on a real batch axis ...
    assert orc.chunk_bounds(3, 128) == [0, 42, 84, 128]
    b = orc.chunk_bounds(7, 128)
    assert [b[i + 1] - b[i] for i in range(7)] == [18] * 6 + [20]
    # ... and on the singleton axis the reference actually hands it: everything goes to the last subset
    assert orc.chunk_bounds(3, 1) == [0, 0, 0, 1]
    assert orc.chunk_bounds(7, 1) == [0] * 7 + [1]


@pytest.mark.parametrize("name", forward_cases())
def test_forward_with_missing_modalities_matches_reference(name):
    """MOE.forward with all / only one modality present (the cross-generation calls of
    models/trainer.py:179-215), the oracle against the reference's own outputs"""
    meta, g = load_golden(name)
    shapes = orc.model_param_shapes(meta["mods"], meta["D"])
    p = gw.make_params(shapes, meta["seed"])
    full = _batch(g)
    for ci, present in enumerate(meta["present"]):
        batch = {k: dict(v, data=v["data"] if i in present else None) for i, (k, v) in enumerate(full.items())}
        eps = [torch.from_numpy(g[f"c{ci}/eps_{i}"]) for i in range(int(g[f"c{ci}/n_eps"]))]
        assert meta["mixing"] != "moe" or len(eps) == len(present)
        if meta["mixing"] == "dmvae":
            with torch.no_grad():
                outs, (j_mu, j_var) = orc.dmvae_forward(p, meta["mods"], batch, eps, meta["D"])
            for m, o in enumerate(outs):
                pre = f"c{ci}/"
                assert (o["q_shared"] is not None) == bool(g[pre + f"has_q_{m}"]) == bool(g[pre + f"has_qp_{m}"])
                if o["q_shared"] is not None:
                    _close(o["q_shared"][0], g[pre + f"q_mu_{m}"], 1e-5, f"c{ci} q_mu_{m}")
                    _close(o["q_shared"][1], g[pre + f"q_sigma_{m}"], 1e-5, f"c{ci} q_sigma_{m}")
                    _close(o["q_private"][0], g[pre + f"qp_mu_{m}"], 1e-5, f"c{ci} qp_mu_{m}")
                    _close(o["q_private"][1], g[pre + f"qp_sigma_{m}"], 1e-5, f"c{ci} qp_sigma_{m}")
                _close(j_mu, g[pre + f"joint_mu_{m}"], 1e-5, f"c{ci} joint_mu")
                _close(j_var, g[pre + f"joint_sigma_{m}"], 1e-5, f"c{ci} joint_sigma")
                _close(o["z_shared"], g[pre + f"z_{m}"], 1e-5, f"c{ci} z_{m}")
                _close(gw.summarize(o["px"], 256), g[pre + f"px_{m}"], 1e-5, f"c{ci} px_{m}")
                _close(gw.summarize(o["joint_px"], 256), g[pre + f"jpx_{m}"], 1e-5, f"c{ci} jpx_{m}")
                for src, loc in o["cross"].items():
                    _close(gw.summarize(loc, 256), g[pre + f"cross_{m}_from_{src}"], 1e-5, f"c{ci} cross_{m}_from_{src}")
            want = {k for k in g if k.startswith(f"c{ci}/cross_")}
            assert want == {f"c{ci}/cross_{m}_from_{src}" for m, o in enumerate(outs) for src in o["cross"]}
            continue
        with torch.no_grad():
            q, z, px, cross = orc.moe_forward(p, meta["mods"], batch, eps, meta["D"])
        for m in range(len(meta["mods"])):
            assert (q[m] is not None) == bool(g[f"c{ci}/has_q_{m}"]) == (m in present)
            if q[m] is not None:
                _close(q[m][0], g[f"c{ci}/q_mu_{m}"], 1e-5, f"c{ci} q_mu_{m}")
                _close(q[m][1], g[f"c{ci}/q_sigma_{m}"], 1e-5, f"c{ci} q_sigma_{m}")
            _close(z[m], g[f"c{ci}/z_{m}"], 1e-5, f"c{ci} z_{m}")
            assert tuple(px[m].shape) == tuple(g[f"c{ci}/px_shape_{m}"])
            _close(gw.summarize(px[m], 256), g[f"c{ci}/px_{m}"], 1e-5, f"c{ci} px_{m}")
        for t, (s_, loc) in cross.items():
            _close(gw.summarize(loc, 256), g[f"c{ci}/cross_{t}_from_{s_}"], 1e-5, f"c{ci} cross_{t}_from_{s_}")
        assert {k for k in g if k.startswith(f"c{ci}/cross_")} == {f"c{ci}/cross_{t}_from_{s_}" for t, (s_, _) in cross.items()}
