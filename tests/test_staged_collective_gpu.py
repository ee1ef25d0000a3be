"""The staged (bucketed) all-reduce of the ResNet-50 model's 102 MB gradient on hardware: one RCCL rank
(`--force-collective` structure: the collectives move nothing but are really launched and captured), the shipped
CdSprites+ config.  Parameters after a few captured steps must be bit-identical to the single-collective step."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json, torch
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch.distributed as dist
from multimodal_vae_comparison_amd import parallel
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
from multimodal_vae_comparison_amd.synthetic import workload
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
desc, cfg, dims, data, meta = workload("cdsprites_shipped", 8, device=dev, seed=1)
out = {}
for staged in ("1", "0"):
    os.environ["MMVAE_DP_STAGED"] = staged
    torch.manual_seed(0)
    DropoutState._next_seed[0] = 0x1234567
    tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
    tr.model.train()
    tr.configure_optimizers()
    parallel.setup_replica(tr, 0, 1)
    tr.capture({k: dict(v) for k, v in data.items()}, 2)      # the N > 1 step structure on one rank
    for _ in range(3):
        tr.fused_step(2)
    torch.cuda.synchronize()
    st = getattr(tr, "_stager", None)
    out[staged] = {"sum": float(tr.flat.data.double().sum()), "abs": float(tr.flat.data.double().abs().sum()),
                   "in_graph": bool(tr._collective_in_graph), "n": st.n_collectives if st is not None else 1}
    torch.save(tr.flat.data.cpu(), sys.argv[3] + staged)
a, b = torch.load(sys.argv[3] + "1"), torch.load(sys.argv[3] + "0")
out["equal"] = bool(torch.equal(a, b))
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_staged_all_reduce_trains_bit_identically(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 300 + 40)
    r = subprocess.run([sys.executable, str(script), ROOT, port, str(tmp_path / "p")], capture_output=True, text=True,
                       timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    out = json.loads(line[0][len("RESULT "):])
    assert out["1"]["in_graph"] and out["0"]["in_graph"], out
    assert out["1"]["n"] >= 3 and out["0"]["n"] == 1, out          # 102 MB in >= 25 MB buckets vs one collective
    assert out["equal"], out
