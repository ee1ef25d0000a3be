"""Where does the run-to-run spread of the cfg2 step time (0.405 - 0.419 ms on one box) come from?  Several timed
segments of one captured trainer, then fresh trainers (new allocations, new capture) in the same process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
for rep in range(4):
    tr, desc, meta = bench._build("cfg2", 128, dev, 0, 1, 1)
    seg = []
    for _ in range(30):
        tr.fused_step(1)
    for s in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300):
            tr.fused_step(1)
        torch.cuda.synchronize()
        seg.append((time.perf_counter() - t0) / 300 * 1e3)
    print(f"trainer {rep}: " + " ".join(f"{v:.4f}" for v in seg) + f"   flat.data @ {tr.flat.data.data_ptr():#x}")
    if rep == 1:
        junk = torch.empty(37 << 20, device=dev)      # shift the following allocations
