"""Start times of the instrumented kernels inside a REAL captured training step (no extra graph nodes, no profiler):
run with the -DMMVAE_TRACE build of the library,
    MMVAE_HIP_LIB=tools/probe/libmmvae_trace.so python tools/probe/trace_step.py
(build: every csrc/*.hip with -DMMVAE_TRACE linked into tools/probe/libmmvae_trace.so)."""
import ctypes
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MMVAE_HIP_LIB", os.path.join(here, "libmmvae_trace.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch

from multimodal_vae_comparison_amd import hipops as H
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import cdsprites_batch, cdsprites_config

NAMES = {0: "img enc conv1", 1: "img enc conv2", 2: "img enc conv3", 3: "img enc conv4",
         4: "img dec convT(4x4)", 5: "img dec convT(8x8)", 6: "img dec convT(16x16)", 7: "img dec convT3 (last fwd)",
         8: "img dec bwd convT3", 9: "img dec bwd convT(16)", 10: "img dec bwd convT(8)", 11: "img dec bwd convT(4)",
         12: "img enc bwd conv4", 13: "img enc bwd conv3", 14: "img enc bwd conv2", 15: "img enc bwd conv1 wgrad",
         16: "txt enc layer fwd", 17: "txt dec layer fwd", 18: "txt enc layer bwd", 19: "txt dec layer bwd",
         20: "adam", 21: "reduce_segments", 22: "poe fwd", 23: "poe bwd",
         24: "  gemm_grouped", 25: "  rgemm16", 26: "  rgemm", 27: "  rgemm_grouped", 28: "  bce_rowsum", 29: "  ce_time_fwd",
         30: "  embed_pe_fwd", 31: "  embed_pe_bwd", 32: "  permute_mask", 33: "  dropout_advance", 34: "  dropout_act",
         35: "  txt_wgrad", 36: "img dec convT3 col2im fwd", 37: "  gemm_b16", 38: "  gemm_b16 linear bwd",
         39: "  rgemm_batch (parked dW)"}
dev = torch.device("cuda", 0)
B = int(os.environ.get("TRACE_BATCH", 128))
torch.manual_seed(0)
tr = MultimodalVAE(cdsprites_config("mopoe", 32, batch_size=B), device=dev)
tr.model.train()
tr.configure_optimizers()
batch = cdsprites_batch(B, 32, seed=1, device=dev)
table = torch.zeros(8 + 2 * 2048, dtype=torch.int64, device=dev)
L = ctypes.CDLL(os.environ["MMVAE_HIP_LIB"])
for m in ("conv", "txtlayer", "txtwave", "optim", "latent", "gemm", "loss", "text", "twgrad"):
    fn = getattr(L, f"mmvae_trace_set_{m}", None)      # (the product build has no stamps: the step time only)
    if fn is None:
        continue
    fn.argtypes = [ctypes.c_void_p]
    assert fn(table.data_ptr()) == 0
tr.capture(batch, 1)
for _ in range(60):
    tr.fused_step(1)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        tr.fused_step(1)
    e1.record()
    torch.cuda.synchronize()
    print(f"step: {e0.elapsed_time(e1) * 1000 / 300:.1f} us")
if int(table[0]) == 0:
    sys.exit(0)
v = table.cpu().tolist()
n_ev = v[0]
ev = [(v[9 + 2 * (k & 2047)], v[8 + 2 * (k & 2047)]) for k in range(max(0, n_ev - 2047), n_ev)]     # (time, id), launch order
ev.sort()
starts = [i for i, (t, k) in enumerate(ev) if k == 0]           # conv1 opens a step
assert len(starts) >= 3, "no complete step in the event log"
step = ev[starts[-2]:starts[-1]]
t0 = step[0][0]
prev = t0
for t, k in step:
    print(f"{(t - t0) / 100.0:8.2f} us  (+{(t - prev) / 100.0:6.2f})  {NAMES.get(k, k)}")
    prev = t
print(f"{(ev[starts[-1]][0] - t0) / 100.0:8.2f} us  next step's conv1   ({len(step)} instrumented launches per step)")
