"""Same-box A/B of the cfg2 step with the fused linear chains (ops.LINEAR_CHAIN) on and off: interleaved rounds of bench.py's own
timed loop.  python tools/probe/chain_instep.py [batch] [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from multimodal_vae_comparison_amd import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    def barrier():
        torch.cuda.synchronize()
    trs = {}
    for fused in (False, True):
        ops.LINEAR_CHAIN = fused
        trs[fused], _, _ = bench._build("cfg2", B, dev, 0, 1, 1)
        print("chain" if fused else "per-layer", "abi calls/step:", trs[fused].abi_calls_in_graph, flush=True)
    for r in range(rounds):
        for fused in (False, True):
            dt, out = bench._timed(trs[fused], 300, 30, 1, barrier)
            print(f"round {r} {'chain    ' if fused else 'per-layer'} {1e3 * dt / 300:.4f} ms/step loss {float(out['loss']):.2f}",
                  flush=True)
    print("timeouts:", ops.chain_timeouts(dev))


if __name__ == "__main__":
    main()
