"""A few launches of each conv kernel at one batch size (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_vae_comparison_amd import hipops as H, ops
B = int(os.environ.get("PMC_B", "4096"))
dev = "cuda"
with torch.no_grad():
    x = torch.randn(B, 32, 32, 32, device=dev); w = torch.randn(32, 32, 4, 4, device=dev) * .05; b = torch.zeros(32, device=dev)
    for _ in range(3): ops.conv2d_k4s2(x, w, b, H.ACT_SILU)
    x2 = torch.randn(B, 32, 16, 16, device=dev)
    for _ in range(3): ops.convT2d_k4s2(x2, w, b, H.ACT_RELU, 0)
    L = H.lib()
    dy = torch.randn(B, 32, 16, 16, device=dev); dw = torch.zeros(32, 32, 4, 4, device=dev); db = torch.zeros(32, device=dev)
    ws = torch.empty(L.mmvae_conv_wgrad_ws_floats(B, 32, 32, 16), device=dev)
    for _ in range(3): L.mmvae_conv2d_k4s2_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 32, 32, 16, 1, 1, H.stream())
    torch.cuda.synchronize()
