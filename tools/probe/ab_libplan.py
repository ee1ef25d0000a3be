"""Same-box A/B of a runtime plan switch of the LIBRARY (an `int f(int)` entry point that returns the previous setting, e.g.
mmvae_rc_patch_plan) inside the real captured step: the two trainers are captured under the two settings, then timed in
interleaved rounds.  python tools/probe/ab_libplan.py FUNC [config] [batch] [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from multimodal_vae_comparison_amd import hipops as H  # noqa: E402


def main():
    fn = getattr(H.lib(), sys.argv[1])
    cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else None
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    trs = {}
    for on in (0, 1):
        fn(on)
        trs[on], _, meta = bench._build(cfg, B, dev, 0, 1, 1)
    steps = 300 if meta["B"] <= 256 and cfg.startswith("cfg") else 60
    for r in range(rounds):
        for on in (0, 1):
            dt, out = bench._timed(trs[on], steps, 20, 1, torch.cuda.synchronize)
            print(f"round {r} {sys.argv[1]}={on} {1e3 * dt / steps:.4f} ms/step loss {float(out['loss']):.2f}", flush=True)


if __name__ == "__main__":
    main()
