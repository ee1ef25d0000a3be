"""How many fold segments does one step of a workload register, and how many would be left if segments whose source
columns and destinations continue the previous one's (same rows, same stride) were merged?"""
import sys
import torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda", 0)
desc, cfg, dims, data, meta = workload(name, None, device=dev, seed=1)
tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
tr.model.train(); tr.configure_optimizers()
segs = []
add0 = ops.GradReducer.add.__func__
def add(cls, src_ptr, dst, rows, length, stride):
    segs.append((src_ptr, dst.data_ptr(), int(rows), int(length), int(stride)))
    return add0(cls, src_ptr, dst, rows, length, stride)
ops.GradReducer.add = classmethod(add)
tr._one = torch.ones((), device=dev)
ops.LincombRows.unit_seed_ptr = tr._one.data_ptr()
tr._fwd_bwd(data); tr._finish_step()
segs.clear()
tr._fwd_bwd(data); tr._finish_step()
torch.cuda.synchronize()
merged = []
for s in segs:
    if merged:
        sp, dp, r, ln, sd = merged[-1]
        if s[0] == sp + 4 * ln and s[1] == dp + 4 * ln and s[2] == r and s[4] == sd:
            merged[-1] = (sp, dp, r, ln + s[3], sd)
            continue
    merged.append(s)
print(name, "segments", len(segs), "-> merged", len(merged))
g0 = tr.flat.grad.data_ptr(); split = tr.flat.split
dec = [s for s in merged if s[1] >= g0 + 4 * split]
print("  in the decoders' + prior's range:", len(dec), " encoders':", len(merged) - len(dec))
import collections
print("  lens of merged:", collections.Counter((s[2], s[3]) for s in merged).most_common(12))
