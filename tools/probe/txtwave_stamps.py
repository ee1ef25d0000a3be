"""s_memtime stamps of one sequence's wave inside csrc/txtwave.hip (-DTV_PROBE build of the library): where the layer's
time goes.  Build: make -C multimodal_vae_comparison_amd/csrc probe_txtwave; run with MMVAE_HIP_LIB=<that .so>."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H
from multimodal_vae_comparison_amd.models import decoders, encoders
from multimodal_vae_comparison_amd.models.nn_modules import DropoutState

DEV, L = "cuda", 32
lib = H.lib()
for N in [int(a) for a in sys.argv[1:]] or [128]:
    for dec, d in ((False, 54), (True, 32)):
        torch.manual_seed(0)
        layer = (decoders.HipTransformerDecoderLayer if dec else encoders.HipTransformerEncoderLayer)(d, 2, 128).to(DEV)
        x = torch.randn(L, N, d, device=DEV, requires_grad=True)
        mem = torch.randn(N, d, device=DEV, requires_grad=True)
        mask = torch.ones(N, L, dtype=torch.uint8, device=DEV)
        st = DropoutState().to(DEV)
        sites = ("attn", "drop1", "xattn", "drop2", "ffn", "drop3") if dec else ("attn", "drop1", "ffn", "drop2")
        ds = {k: st.spec(0, 0, i + 1, 0.1, k) for i, k in enumerate(sites)}
        for _ in range(3):
            out = layer(x, mem, mask, ds) if dec else layer(x, mask, ds)
            out.backward(torch.ones_like(out))
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 64)()
        assert lib.mmvae_txt_wave_stamps(buf) == 0
        for direction, name in ((0, "fwd"), (1, "bwd")):
            t = [buf[32 * direction + i] for i in range(32)]
            if t[0] == 0:
                continue
            pts = [(i, t[i] - t[0]) for i in range(32) if t[i]]
            pts.sort(key=lambda p: p[1])
            print(f"N {N} {'dec' if dec else 'enc'} {name}: " + "  ".join(f"[{i}] {v}" for i, v in pts) + "  (shader cycles)")
