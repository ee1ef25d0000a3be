"""ELBO-curve parity (SURVEY 8(d) "Parity procedure", BASELINE north_star "ELBO curves overlapping the reference within
tolerance"): N train-mode steps of BASELINE configs[1] (MoPoE, CdSprites+ shapes, B=128, T=32, D=32, dropout ON,
Adam(amsgrad)) on the HIP path and on the CPU oracle, several seeds each, SAME initial parameters per seed index and
the SAME batch sequence, independent noise / dropout streams (the HIP kernels' counter-based generators vs torch's
CPU generator -- bit parity under dropout is pinned separately with extracted masks, tests/test_parity_e2e.py).

    python tools/curve_parity.py [--steps 200] [--seeds 3] [--lr 1e-3] [--out profiles/r02_curve_parity.json]

Reports, per path, the moving-average loss / kld / reconstruction terms per seed, the seed envelope of the oracle and
whether the HIP seed-mean lies inside it (reference training loop: models/trainer.py:117-128 + Adam(amsgrad) :79-81).
The oracle is test infrastructure: this tool is a checker, nothing here is shipped or timed.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multimodal_vae_comparison_amd.synthetic import CD_MODS, cdsprites_batch, cdsprites_config  # noqa: E402


def make_data(n_batches, B, T, seed=100):
    """a small fixed 'dataset': n_batches batches cycled in order (both paths see the same sequence)"""
    return [cdsprites_batch(B, T, seed=seed + i) for i in range(n_batches)]


def init_params(seed, D):
    """torch-default initial parameters of the HIP model for `seed`, as CPU tensors keyed like the reference"""
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    torch.manual_seed(seed)
    tr = MultimodalVAE(cdsprites_config("mopoe", D), device="cpu")
    return {k: p.detach().clone() for k, p in tr.model.named_parameters() if p.requires_grad}


def run_hip(seed, data, steps, lr, D, device="cuda", captured=True):
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    params = init_params(seed, D)
    torch.manual_seed(1000 + seed)            # noise / dropout seeds of this run
    tr = MultimodalVAE(cdsprites_config("mopoe", D, lr=lr), device=device)
    named = dict(tr.model.named_parameters())
    with torch.no_grad():
        for k, v in params.items():
            named[k].copy_(v.to(device))
    tr.model.train()
    tr.configure_optimizers()
    dev_data = [{k: {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in b.items()}
                for b in data]
    hist = []
    if captured:
        tr.capture({k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                    for k, v in dev_data[0].items()}, 1)
    for i in range(steps):
        b = dev_data[i % len(dev_data)]
        if captured:
            tr.load_batch(b)
            out = tr.fused_step(1)
        else:
            out = tr.model.objective(b)
            out["loss"].backward()
            tr.optimizer.step()
        hist.append([float(out["loss"]), float(out["kld"])] + [float(r.mean()) for r in out["reconstruction_loss"]])
    return hist


def run_oracle(seed, data, steps, lr, D):
    from oracle import mmvae_oracle as orc
    params = {k: v.clone().requires_grad_(True) for k, v in init_params(seed, D).items()}
    torch.manual_seed(2000 + seed)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    hist = []
    B = data[0]["mod_1"]["data"].shape[0]
    for i in range(steps):
        b = data[i % len(data)]
        eps = [torch.randn(1, B, D) for _ in range(2)]
        out = orc.mopoe_objective(params, CD_MODS, b, eps, D, train=True)
        out["loss"].backward()
        with torch.no_grad():
            orc.adam_amsgrad_step(params, {k: p.grad for k, p in params.items()}, state, lr, i + 1)
            for p in params.values():
                p.grad = None
        hist.append([float(out["loss"]), float(out["kld"])] + [float(r.mean()) for r in out["reconstruction_loss"]])
    return hist


def moving_average(x, w):
    out, acc = [], 0.0
    for i, v in enumerate(x):
        acc += v
        if i >= w:
            acc -= x[i - w]
        out.append(acc / min(i + 1, w))
    return out


def compare(hip, orc_, window=10, skip=5):
    """hip / orc_: [seed][step][quantity].  Per quantity: the HIP seed-mean of the moving average against the oracle's
    seed envelope [min, max] widened by half its own width on each side (3 seeds under-sample the spread); returns the
    fraction of steps inside and the largest relative excursion outside."""
    names = ["loss", "kld", "recon_mod_1", "recon_mod_2"]
    res = {}
    for q, name in enumerate(names):
        h = [moving_average([s[q] for s in run], window) for run in hip]
        o = [moving_average([s[q] for s in run], window) for run in orc_]
        steps = len(h[0])
        inside, worst = 0, 0.0
        for t in range(skip, steps):
            hm = sum(r[t] for r in h) / len(h)
            lo, hi = min(r[t] for r in o), max(r[t] for r in o)
            om = sum(r[t] for r in o) / len(o)
            pad = 0.5 * (hi - lo) + 1e-3 * abs(om)
            if lo - pad <= hm <= hi + pad:
                inside += 1
            else:
                worst = max(worst, min(abs(hm - lo), abs(hm - hi)) / max(abs(om), 1e-12))
        res[name] = {"fraction_inside": inside / max(1, steps - skip), "worst_relative_excursion": worst,
                     "hip_mean_first": sum(r[skip] for r in h) / len(h), "hip_mean_last": sum(r[-1] for r in h) / len(h),
                     "oracle_mean_first": sum(r[skip] for r in o) / len(o), "oracle_mean_last": sum(r[-1] for r in o) / len(o),
                     "oracle_seed_spread_last": max(r[-1] for r in o) - min(r[-1] for r in o)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--seq", type=int, default=32)
    ap.add_argument("--latents", type=int, default=32)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_curve_parity.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    data = make_data(a.batches, a.batch, a.seq)
    hip = [run_hip(s, data, a.steps, a.lr, a.latents) for s in range(a.seeds)]
    orc_ = [run_oracle(s, data, a.steps, a.lr, a.latents) for s in range(a.seeds)]
    res = compare(hip, orc_)
    window = 10
    out = {"workload": f"configs[1] MoPoE CdSprites+ shapes, B={a.batch}, T={a.seq}, D={a.latents}, train mode (dropout "
                       f"0.1), Adam(amsgrad) lr {a.lr}, {a.steps} steps over {a.batches} cycled synthetic batches, "
                       f"{a.seeds} seeds per path",
           "summary": res, "moving_average_window": window,
           "hip_loss_ma": [[round(v, 3) for v in moving_average([s[0] for s in run], window)][::5] for run in hip],
           "oracle_loss_ma": [[round(v, 3) for v in moving_average([s[0] for s in run], window)][::5] for run in orc_],
           "hip_kld_ma": [[round(v, 4) for v in moving_average([s[1] for s in run], window)][::5] for run in hip],
           "oracle_kld_ma": [[round(v, 4) for v in moving_average([s[1] for s in run], window)][::5] for run in orc_]}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
