"""Stand-alone time of one fused text layer (forward, forward + backward) through the model's own modules, for the
encoder (d = 54) and decoder (d = 32) layers at N sequences: run under MMVAE_TXT_WAVE=0 / 1 / 3 to compare the
workgroup-per-sequence kernels (csrc/txtlayer.hip) with the wave-per-sequence ones (csrc/txtwave.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd.models import decoders, encoders
from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
from tools.gtime import timeit

DEV = "cuda"
L = 32
for N in [int(a) for a in sys.argv[1:]] or [128, 1000]:
    for dec, d in ((False, 54), (True, 32)):
        torch.manual_seed(0)
        layer = (decoders.HipTransformerDecoderLayer if dec else encoders.HipTransformerEncoderLayer)(d, 2, 128).to(DEV)
        x = torch.randn(L, N, d, device=DEV)
        mem = torch.randn(N, d, device=DEV)
        mask = torch.ones(N, L, dtype=torch.uint8, device=DEV)
        dy = torch.randn(L, N, d, device=DEV)
        st = DropoutState().to(DEV)
        sites = ("attn", "drop1", "xattn", "drop2", "ffn", "drop3") if dec else ("attn", "drop1", "ffn", "drop2")
        ds = {k: st.spec(0, 0, i + 1, 0.1, k) for i, k in enumerate(sites)}

        def fwd():
            with torch.no_grad():
                return layer(x, mem, mask, ds) if dec else layer(x, mask, ds)

        xg, mg = x.clone().requires_grad_(True), mem.clone().requires_grad_(True)

        def both():
            out = layer(xg, mg, mask, ds) if dec else layer(xg, mask, ds)
            out.backward(dy)

        tf = timeit(fwd, reps=10, n=10)
        try:
            tb = timeit(both, reps=10, n=5)
        except Exception as e:       # autograd inside a capture may be refused: eager timing
            tb = float("nan")
        print(f"N {N:5d} {'dec d=32' if dec else 'enc d=54'}: forward {tf:7.2f} us, forward+backward(+wgrad) {tb:7.2f} us  "
              f"[MMVAE_TXT_WAVE={os.environ.get('MMVAE_TXT_WAVE', 'default')}]")
