"""bench.py -- training samples/sec of the MoPoE CdSprites+ level-2 step (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + backward + Adam(amsgrad) of MoPoE (CNN2 image tower + TxtTransformer text tower,
n_latents 32) on one synthetic batch of 128 samples per GPU (64x64x3 image + 32-token text), inputs resident in
HBM.  N > 1: one process per GPU, weak scaling (128 samples per GPU), ONE RCCL all-reduce of the flat 3.95 MB
gradient buffer per step.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

# rank 0 prints ONE JSON line on stdout.  RCCL writes its debug output (the version banner of NCCL_DEBUG=VERSION, which
# the GPU boxes export, and its warnings) to stdout as well, unbuffered and in the middle of other lines: send it to a
# file instead
os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_debug.%h.%p.log")
if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":      # the banner ignores NCCL_DEBUG_FILE
    del os.environ["NCCL_DEBUG"]

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 98.9e6        # 16 477 056 MAC fwd x 2 x 3 (fwd + dgrad + wgrad)
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--batch", type=int, default=128, help="samples per GPU (BASELINE configs[1]: 128)")
    p.add_argument("--seq", type=int, default=32)
    p.add_argument("--latents", type=int, default=32)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--force-collective", action="store_true",
                   help="one GPU, but the multi-GPU step structure (graph, RCCL all-reduce over 1 rank, Adam launch)")
    return p.parse_args()


def cpu_baseline(B, T, D, budget_s):
    """The CPU restatement of the same step (oracle, kind "port"): objective + backward + Adam(amsgrad), train mode
    (dropout on, as the reference trains), all host cores, on a bounded number of steps."""
    from oracle import golden_weights as gw
    from oracle import mmvae_oracle as orc
    from multimodal_vae_comparison_amd.synthetic import cdsprites_batch
    mods = [{"enc": "CNN2", "dec": "CNN", "data_dim": [64, 64, 3], "ltype": "bce"},
            {"enc": "TxtTransformer", "dec": "TxtTransformer", "data_dim": [45, 27, 1], "ltype": "category_ce"}]
    params = gw.make_params(orc.model_param_shapes(mods, D), 0, requires_grad=True)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    batch = cdsprites_batch(B, T, seed=1)

    def step(i):
        eps = [torch.randn(1, B, D) for _ in range(2)]
        out = orc.mopoe_objective(params, mods, batch, eps, D, train=True)
        out["loss"].backward()
        with torch.no_grad():
            orc.adam_amsgrad_step(params, {k: p.grad for k, p in params.items()}, state, 1e-4, i)
            for p in params.values():
                p.grad = None

    # torch's intra-op pool does not scale to hundreds of threads on ops this small: calibrate the thread count
    # (a few steps each) and time the baseline at the fastest setting.
    ncpu = os.cpu_count() or 1
    best = (float("inf"), 1)
    it = 0
    for nt in sorted({t for t in (4, 8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(nt)
        it += 1
        step(it)
        t0 = time.perf_counter()
        for _ in range(2):
            it += 1
            step(it)
        dt = (time.perf_counter() - t0) / 2
        if dt < best[0]:
            best = (dt, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < 400:
        n += 1
        step(it + n)
    dt = time.perf_counter() - t0
    return {"value": round(n * B / dt, 1), "unit": "samples/s", "cores": cores, "host_cpus": ncpu, "kind": "port",
            "sample": f"{n} steps of the oracle (oracle/mmvae_oracle.py, train mode) at B={B}, T={T}, D={D}, "
                      f"{1e3 * dt / n:.1f} ms/step"}


# HBM bytes per launch of the conv2 forward kernel at B=128 from the PMC passes of tools/gpu_pmc.sh
# (2 * FETCH_SIZE 10059.9 KB + WRITE_SIZE 4096.0 KB); bench.py cannot collect counters itself.
CONV2_FWD_HBM_BYTES_B128 = int((2 * 10059.9 + 4096.0) * 1024)


def dominant_kernel_roofline(B, device):
    """Average duration of the dominant kernel at this workload's shape, measured live with HIP events on the
    launch stream: the 32->32 channel 4x4/s2 gather conv at 32x32 -> 16x16 (encoder conv2; the decoder's
    convT2 backward-data is the same kernel and shape).  Algorithmic FLOPs per launch = 2 * B*16*16*32 * 512."""
    from multimodal_vae_comparison_amd import ops
    from multimodal_vae_comparison_amd import hipops as H
    x = torch.randn(B, 32, 32, 32, device=device)
    w = torch.randn(32, 32, 4, 4, device=device) * 0.05
    b = torch.zeros(32, device=device)
    with torch.no_grad():
        for _ in range(5):
            ops.conv2d_k4s2(x, w, b, H.ACT_SILU)
        torch.cuda.synchronize()
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d_k4s2(x, w, b, H.ACT_SILU)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * B * 16 * 16 * 32 * 512
    ach = flops / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "conv_gather_kernel<32,*> (conv2 fwd shape)", "achieved": round(ach, 2),
            "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            "avg_us": round(us, 2), "traffic": CONV2_FWD_HBM_BYTES_B128 if B == 128 else None,
            "traffic_unit": "bytes/launch",
            "traffic_source": "profiles/r01_e_pmc_conv_b128.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                              "FETCH doubled per the gfx950 16-B/lane correction); algorithmic 21.0e6"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == a.gpus or world == 1 and a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    # --force-collective (testing on one GPU): take the multi-GPU structure of the step -- graph without the optimiser,
    # one RCCL all-reduce of the flat gradients, separate Adam launch -- with a single-rank process group
    path_world = world
    if a.force_collective and world == 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1, device_id=dev)
        path_world = 2

    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    from multimodal_vae_comparison_amd.synthetic import cdsprites_batch, cdsprites_config

    torch.manual_seed(0)                                   # identical replicas on every rank
    tr = MultimodalVAE(cdsprites_config("mopoe", a.latents, batch_size=a.batch), device=dev)
    tr.model.train()
    opt = tr.configure_optimizers()
    if world > 1:
        opt.grad_scale = 1.0 / world                       # all-reduce(sum) then average inside the Adam kernel
    batch = cdsprites_batch(a.batch, a.seq, seed=1 + rank, device=dev)
    tr.capture(batch, path_world)      # world 1: the Adam step is part of the captured graph

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.fused_step(path_world)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.fused_step(path_world)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].item())

    if rank == 0:
        sps = a.steps * a.batch * world / dt
        res = {"metric": "training samples/sec, MoPoE CdSprites+ L2 (fwd+bwd+Adam)", "value": round(sps, 1),
               "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[1]: MoPoE, CdSprites+ L2 shapes, CNN2 image tower + TxtTransformer "
                                      f"text tower, n_latents={a.latents}, batch={a.batch}/GPU, T={a.seq}, "
                                      "Adam(amsgrad) lr 1e-4, beta 1",
                          "global_batch": a.batch * world, "parallelism": f"dp{world}",
                          "step_flops_fraction_of_f32_mfma_peak": round(
                              sps / world * FLOP_PER_SAMPLE / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)},
               "final_loss": round(loss, 3)}
        if world == 1:
            res["roofline"] = dominant_kernel_roofline(a.batch, dev)
            if not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(a.batch, a.seq, a.latents, a.cpu_seconds)
                res["config"]["gpu_over_cpu"] = round(sps / res["cpu_baseline"]["value"], 1)
    if world > 1 or path_world > 1:
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
