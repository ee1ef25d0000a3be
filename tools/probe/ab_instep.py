"""Same-box A/B of a module switch of `ops` (e.g. CONVT3_BCE) inside the real captured step: interleaved rounds of bench.py's
own timed loop.  python tools/probe/ab_instep.py SWITCH [config] [batch] [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from multimodal_vae_comparison_amd import ops  # noqa: E402


def main():
    switch = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else None
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    assert hasattr(ops, switch), switch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    def barrier():
        torch.cuda.synchronize()
    trs = {}
    for kv in sys.argv[5:]:      # further switches held fixed: NAME=0|1
        k, v = kv.split("=")
        setattr(ops, k, bool(int(v)))
    for on in (False, True):
        setattr(ops, switch, on)
        trs[on], _, meta = bench._build(cfg, B, dev, 0, 1, 1)
        print(f"{switch}={on} abi calls/step:", trs[on].abi_calls_in_graph, flush=True)
    setattr(ops, switch, True)
    steps = 300 if meta["B"] <= 256 else 60
    for r in range(rounds):
        for on in (False, True):
            dt, out = bench._timed(trs[on], steps, 30, 1, barrier)
            print(f"round {r} {switch}={on!s:5} {1e3 * dt / steps:.4f} ms/step loss {float(out['loss']):.2f}", flush=True)


if __name__ == "__main__":
    main()
