"""bench.py -- training samples/sec of the multimodal-VAE step on N MI355X (default: BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 200 --warmup 20 [--config cfg1|cfg2|cfg3|cfg4|cfg5|mnistsvhn]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + backward + Adam(amsgrad) of the mixer on one synthetic batch per GPU, inputs resident in HBM.
Default workload cfg2 = MoPoE (CNN2 image tower + TxtTransformer text tower, n_latents 32), 128 samples per GPU
(64x64x3 image + 32-token text).  N > 1: one process per GPU, weak scaling, ONE RCCL all-reduce of the flat gradient
buffer per step (multimodal_vae_comparison_amd/parallel.py).  Prints ONE JSON line (rank 0).
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

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 at 32 cycles: 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz (dense)
METRIC = {"cfg2": "training samples/sec, MoPoE CdSprites+ L2 (fwd+bwd+Adam)"}


SETTLE_STEPS = 40      # untimed set-up replays before the warm-up (see main)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--config", default="cfg2", help="BASELINE workload: cfg1..cfg5 | mnistsvhn (default cfg2, the one "
                                                    "the metric is quoted on)")
    p.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the workload's)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the non-headline figures (input pipeline, large batch)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="process-group backend for N > 1: nccl = RCCL over xGMI (one GPU per rank); gloo moves the CUDA "
                        "buffers through the host and lets several ranks share one GPU (N > 1 readiness test on a "
                        "1-GPU box: LOCAL_RANK is taken modulo the visible device count)")
    p.add_argument("--settle", type=int, default=SETTLE_STEPS,
                   help="untimed set-up replays between the first window (W warm-ups + K timed steps right after the capture, "
                        "reported as first_window_ms_per_step) and the headline region (W warm-ups + K timed steps); "
                        "0 = the headline IS the first window")
    p.add_argument("--force-collective", action="store_true",
                   help="one GPU, but the multi-GPU step structure (graph, RCCL all-reduce over 1 rank, Adam launch)")
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port") timed on this box's host cores, SURVEY 8(d)
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo, limited to the CPUs this process may run on"""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if core is not None:
                        seen.add((phys, core))
                    phys = core = None
        n = len(seen) or allowed
    except OSError:
        n = allowed
    return max(1, min(n, allowed))


def _oracle_step_fn(meta, batch):
    """one optimisation step of the oracle for this workload: objective + backward + Adam(amsgrad), train mode (dropout
    on, as the reference trains)"""
    from oracle import golden_weights as gw
    from oracle import mmvae_oracle as orc
    mods, D, B, mixing = meta["mods"], meta["D"], meta["B"], meta["mixing"]
    K = meta.get("K", 1)
    params = gw.make_params(orc.model_param_shapes(mods, D), 0, requires_grad=True)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    M = len(mods)
    if mixing == "dmvae":
        P = [m.get("private") for m in mods]
        shapes = [D] + [d for i in range(M) for d in [D, P[i]] + [D] * (M - 1)]
    elif mixing == "poe":
        shapes = [D] * (2 ** M - 1)
    else:
        shapes = [D] * M
    kw = {k: meta[k] for k in ("obj", "K", "prior") if k in meta}

    def step(i):
        eps = [torch.randn(K, B, d) for d in shapes]
        out = orc.OBJECTIVES[mixing](params, mods, batch, eps, D, train=True, **kw)
        out["loss"].backward()
        with torch.no_grad():
            grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
            orc.adam_amsgrad_step(params, grads, state, 1e-4, i)
            for p in params.values():
                p.grad = None
    return step


def _time_steps(step, warm, timed, budget_s, it0):
    it = it0
    t0 = time.perf_counter()
    for w in range(warm):
        if w >= 1 and time.perf_counter() - t0 > budget_s / 3:      # slow workloads: the warm-up is time-bounded too
            break
        it += 1
        step(it)
    t0 = time.perf_counter()
    n = 0
    while n < timed and (n < 3 or time.perf_counter() - t0 < budget_s):
        n += 1
        it += 1
        step(it)
    return (time.perf_counter() - t0) / n, n, it


def cpu_baseline(meta, batch_cpu):
    """the CPU restatement of the same step at (1) all physical cores, (2) the thread count torch's
    intra-op pool is fastest at on this box (ops this small stop scaling long before 128 threads: measured, all 128
    physical cores of the GPU box run this step 5x SLOWER than one thread), (3) one thread; 10 warm-up + 50 timed
    steps where the time bound allows (each leg is cut at ~12 s, >= 3 timed steps).  `value` / `cores` = the fastest
    of the three (the baseline a CPU user would run); all three legs are reported."""
    B = meta["B"]
    step = _oracle_step_fn(meta, batch_cpu)
    phys = _physical_cores()
    legs = {}
    it = 0
    torch.set_num_threads(phys)
    dt, n, it = _time_steps(step, 10, 50, 12.0, it)
    legs["all_physical_cores"] = (dt, n, phys)
    best = None
    for nt in sorted({t for t in (8, 16, 32) if t < phys}):
        torch.set_num_threads(nt)
        dtc, _, it = _time_steps(step, 1, 3, 2.0, it)
        if best is None or dtc < best[0]:
            best = (dtc, nt)
    if best is not None and best[0] < legs["all_physical_cores"][0]:
        torch.set_num_threads(best[1])
        dt, n, it = _time_steps(step, 3, 50, 10.0, it)
        legs["tuned"] = (dt, n, best[1])
    torch.set_num_threads(1)
    dt, n, it = _time_steps(step, 1, 10, 8.0, it)
    legs["single_thread"] = (dt, n, 1)
    torch.set_num_threads(phys)
    fmt = lambda leg: {"value": round(B / leg[0], 1), "cores": leg[2], "ms_per_step": round(1e3 * leg[0], 1),
                       "timed_steps": leg[1]}
    name, main = min(legs.items(), key=lambda kv: kv[1][0])
    out = {"value": round(B / main[0], 1), "unit": "samples/s", "cores": main[2], "kind": "port",
           "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "physical_cores": phys,
           "sample": f"{main[1]} timed steps of the oracle (oracle/mmvae_oracle.py, train mode, {meta['mixing']}) at "
                     f"B={B}, {1e3 * main[0]:.1f} ms/step on {main[2]} threads (fastest leg: {name})",
           "all_physical_cores": fmt(legs["all_physical_cores"]), "single_thread": fmt(legs["single_thread"])}
    if "tuned" in legs:
        out["tuned"] = fmt(legs["tuned"])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# ---------------------------------------------------------------------------------------------------------------------
def _pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a conv kernel at B=128 from the newest committed PMC summary (tools/gpu_pmc.sh ->
    profiles/r*_pmc_conv_b128.txt): 2 x FETCH_SIZE (the gfx950 16-B/lane correction) + WRITE_SIZE, in KB there."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv_b128.txt")))
    if not files:
        return None, None
    path = files[-1]
    fetch = write = None
    for line in open(path):
        if kernel_prefix not in line:
            continue
        m = re.search(r"'FETCH_SIZE': ([0-9.]+)", line)
        if m and line.startswith("fetch"):
            fetch = float(m.group(1))
        m = re.search(r"'WRITE_SIZE': ([0-9.]+)", line)
        if m and line.startswith("write"):
            write = float(m.group(1))
    if fetch is None or write is None:
        return None, os.path.relpath(path, ROOT)
    return int((2 * fetch + write) * 1024), os.path.relpath(path, ROOT)


def _kernel_flops(name, B, T=32):
    """algorithmic FLOPs of one launch of a cfg2-step kernel at per-GPU batch B (None: not a matrix kernel / unknown):
    the FLOP model behind `roofline.dominant_by_time.frac`"""
    import re
    conv2 = 2.0 * B * 16 * 16 * 32 * 512                     # conv2 / convT(16) shape: one pass (fwd, dgrad or wgrad)
    m = re.search(r"tl::Geom<(\d+), (\d+), (\d+), (true|false)>|tv::Geom<(\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        g = [x for x in m.groups() if x is not None]
        D, FF, NH, dec = int(g[0]), int(g[1]), int(g[2]), g[3] == "true"
        fwd = T * D * 3 * D + 2 * T * T * D + T * D * D + 2 * T * D * FF + (T * D * D + D * D if dec else 0)   # MAC / sequence
        bwd = fwd + 2 * T * T * D                             # data-gradient chain + the recomputed scores
        return 2.0 * B * (bwd if "bwd" in name else fwd)
    if ("conv2d_bwd_fused_kernel<ScatterGeom<4," in name or "convT_bwd_fused_kernel<GatherGeom<32, 5," in name or
            "convT_bwd_fused_b16_kernel<GatherB16Geom<32, 5," in name or
            "conv2d_bwd_fused_b16_kernel<ScatterB16Geom<32, 4>" in name):
        return 2 * conv2
    if ("conv_gather_kernel<GatherGeom<32, 5," in name or "conv_scatter_kernel<ScatterGeom<4," in name or
            "conv_gather_b16p_kernel<GatherB16Geom<32, 5," in name or
            "conv_scatter_b16_kernel<ScatterB16Geom<32, 4>" in name):
        return conv2
    if "txt_wgrad_kernel" in name:                            # both layers' launches averaged: (enc + dec) / 2
        M = T * B
        enc = 2.0 * M * (162 * 54 + 54 * 54 + 128 * 54 + 54 * 128)
        dec = 2.0 * M * (96 * 32 + 32 * 32 + 128 * 32 + 32 * 128 + 32 * 32)
        return (enc + dec) / 2
    if "rgemm_kernel<16, false, false>" in name:              # (round 3: a text layer's (T*B, 162) x (T*B, 54) weight gradient)
        return 2.0 * T * B * 162 * 54
    if "rgemm_grouped_kernel<64, 16, " in name:               # Linear(512, 512) backward: dX + dW
        return 2 * 2.0 * B * 512 * 512
    if "rgemm16_kernel<64, true, true>" in name:
        return 2.0 * B * 512 * 512
    return None


def _dominant_by_time(B):
    """the kernel with the largest share of GPU time in the newest committed rocprofv3 kernel-stats file of this workload
    (profiles/r*_cfg2_b<B>_kernel_stats.csv, made by tools/gpu_evidence_r06.sh from `rocprofv3 --kernel-trace --stats --
    python3 bench.py ...`), and its fraction of the fp32 MFMA peak from the FLOP model above -- the conv2-forward entry of
    `roofline` is the best-tuned kernel, not the one the step spends most of its time in"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_cfg2_b{B}_kernel_stats.csv")))
    if not files:
        return None
    rows = [r for r in csv.DictReader(open(files[-1])) if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
    if not rows:
        return None
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    fl = _kernel_flops(top["Name"], B)
    avg_us = float(top["AverageNs"]) / 1e3
    ach = fl / (avg_us * 1e-6) / 1e12 if fl else None
    return {"kernel": top["Name"].replace("void ", "")[:96], "share": round(float(top["TotalDurationNs"]) / tot, 4),
            "avg_us": round(avg_us, 2), "achieved": round(ach, 2) if ach else None,
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4) if ach else None,
            "source": os.path.relpath(files[-1], ROOT)}


def _event_time_us(fn, reps=50):
    with torch.no_grad():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def dominant_kernel_roofline(meta, device):
    """Average duration of the workload's dominant kernel at the shape it runs at, measured live with HIP events on the
    launch stream (torch's current stream is the stream the C-ABI launches on).
      CdSprites+ towers: the 32->32 channel 4x4/s2 gather conv at 32x32 -> 16x16 (encoder conv2; the decoder's convT2
        backward-data is the same kernel and shape): 2 * B*16*16*32 * 512 FLOP per launch;
      MNIST / SVHN towers: Dec_SVHN's ConvTranspose2d 64->32 at 8x8 -> 16x16 (scatter form), N = the decoder's batch
        (B for DMVAE / MoPoE, M*K*B latent samples for MoE dreg): 2 * N*8*8*64 * 32*16 FLOP per launch."""
    from multimodal_vae_comparison_amd import ops
    from multimodal_vae_comparison_amd import hipops as H
    B = meta["B"]
    if meta["mods"][0]["enc"] == "CNN":
        # ResNet-50 tower: the 3x3 convolution of a layer2 bottleneck (128 -> 128 channels on 8x8 maps of a 64x64 input)
        # with its BatchNorm statistics, on the fused engine (rconv.py): 2 * B*64 * 128 * 128*9 FLOP per launch
        from multimodal_vae_comparison_amd import rconv
        from multimodal_vae_comparison_amd.models.resnet import BatchNorm2d, ConvW
        conv, bn, bnp = ConvW(128, 128, 3, 1, 1, channels_last=True).to(device), BatchNorm2d(128).to(device), BatchNorm2d(128).to(device)
        u, up = rconv.Unit(conv, bn), rconv.Unit(ConvW(64, 128, 1, 1, 0).to(device), bnp)
        M = B * 64
        x = torch.randn(M, 128, device=device)
        bp = up.buffers(M, device)
        bp["mean"].zero_(); bp["sc"].fill_(1.0); bp["rstd"].fill_(1.0)
        us = _event_time_us(lambda: rconv._fwd(u, x, M, M, rconv.PRE_BN_RELU, (bp, bnp.bias), (8, 8, 3, 1, 1), False), reps=30)
        flops = 2.0 * M * 128 * 128 * 9
        ach = flops / (us * 1e-6) / 1e12
        return {"bound": "mfma", "kernel": f"rc_fwd_kernel (layer2 3x3 convolution + BatchNorm statistics, {M} x 128 <- 1152)",
                "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "avg_us": round(us, 2), "traffic": None,
                "traffic_unit": "bytes/launch", "traffic_source": None,
                "note": "eager launches back to back (includes the host launch gap when the kernel is shorter); at batch 24 "
                        "the launch is a chain of hand-over steps, not arithmetic (DESIGN 5c)"}
    if meta["mods"][0]["enc"] == "MNIST":
        N = B * len(meta["mods"]) * meta.get("K", 1) if meta.get("obj") == "dreg" else B
        x = torch.randn(N, 64, 8, 8, device=device)
        w = torch.randn(64, 32, 4, 4, device=device) * 0.05
        b = torch.zeros(32, device=device)
        us = _event_time_us(lambda: ops.convT2d(x, w, b, 2, 1, H.ACT_RELU))
        flops = 2.0 * N * 64 * 64 * 32 * 16
        # (round 4: from 1024 position tiles -- N = 512 -- on this shape runs on the split-bf16 scatter body, conv_scatter_b16.inc)
        kernel = (f"conv_scatter_b16_kernel<ScatterB16Geom<64,3>> (Dec_SVHN conv3 fwd, N={N}; split-bf16 MFMAs)" if N * 2 >= 1024
                  else f"conv_scatter_kernel<ScatterGeom<3,*,*,64>> (Dec_SVHN conv3 fwd, N={N})")
        traffic, src = None, None
    elif meta.get("K", 1) > 1:
        # K-sample MoE on the CdSprites+ towers: both decoders decode M*K*B latent samples, the image decoder's
        # ConvTranspose2d 32->32 at 16x16 -> 32x32 (scatter form) is the largest launch: 2 * N*16*16*32 * 32*16 FLOP
        N = B * len(meta["mods"]) * meta["K"]
        x = torch.randn(N, 32, 16, 16, device=device)
        w = torch.randn(32, 32, 4, 4, device=device) * 0.05
        b = torch.zeros(32, device=device)
        us = _event_time_us(lambda: ops.convT2d_k4s2(x, w, b, H.ACT_RELU, 0), reps=20)
        flops = 2.0 * N * 16 * 16 * 32 * 512
        kernel = (f"conv_scatter_b16_kernel<ScatterB16Geom<32,4>> (Dec_CNN convT2 fwd, N={N}; split-bf16 MFMAs)" if N * 8 >= 512
                  else f"conv_scatter_kernel<ScatterGeom<4,*,*,32>> (Dec_CNN convT2 fwd, N={N})")
        traffic, src = None, None
    else:
        x = torch.randn(B, 32, 32, 32, device=device)
        w = torch.randn(32, 32, 4, 4, device=device) * 0.05
        b = torch.zeros(32, device=device)
        us = _event_time_us(lambda: ops.conv2d_k4s2(x, w, b, H.ACT_SILU))
        flops = 2.0 * B * 16 * 16 * 32 * 512
        # (round 4: this shape runs on the split-bf16 gather kernel from 512 pixel tiles -- batch 64 -- on; fp32 in,
        # fp32 accumulate, fp32 out, six v_mfma_f32_32x32x16_bf16 per 16-deep step: csrc/conv_gather_b16.inc)
        b16 = B * 8 >= 512
        kernel = ("conv_gather_b16p_kernel<GatherB16Geom<32,5,4,4>> (conv2 fwd shape; split-bf16 MFMAs)" if b16
                  else "conv_gather_kernel<32,*> (conv2 fwd shape)")
        traffic, src = _pmc_traffic("conv_gather_b16p_kernel<GatherB16Geom<32, 5" if b16 else "conv_gather_kernel<GatherGeom<32, 5")
        if traffic is None:
            traffic, src = _pmc_traffic("conv_gather_kernel<GatherGeom<32, 5")
        if B != 128:
            traffic = None
    ach = flops / (us * 1e-6) / 1e12
    out = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
           "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "avg_us": round(us, 2),
           "traffic": traffic, "traffic_unit": "bytes/launch",
           "traffic_source": (f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 "
                              f"16-B/lane correction); algorithmic 21.0e6 at B=128") if src else None}
    if "split-bf16" in kernel:
        # `peak` stays the fp32 MFMA peak (the arithmetic the path promises and the reference's matrix rate on this part);
        # the instructions the kernel actually issues have their own ceiling: dense bf16 peak / 6 MFMAs per fp32 product
        out["peak_as_issued"] = round(PEAK_BF16_MFMA_TFLOPS / 6, 1)
        out["frac_as_issued"] = round(ach / (PEAK_BF16_MFMA_TFLOPS / 6), 4)
        out["note"] = ("fp32 operands as three exact bf16 terms, six bf16 MFMAs per product, fp32 accumulate: error vs fp64 "
                       "3e-7 of the tensor maximum, as the fp32-MFMA kernel (tools/probe/gather_b16.py, scatter_b16.py)")
    if meta["mixing"] == "mopoe" and len(meta["mods"]) == 2 and meta["mods"][1]["enc"] == "TxtTransformer":    # cfg2
        out["dominant_by_time"] = _dominant_by_time(B)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def _timed(tr, steps, warmup, path_world, barrier, pre=None):
    for _ in range(warmup):
        if pre:
            pre()
        tr.fused_step(path_world)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if pre:
            pre()
        out = tr.fused_step(path_world)
    barrier()
    return time.perf_counter() - t0, out


def _build(name, batch, dev, rank, world, path_world, input_ring=None):
    from multimodal_vae_comparison_amd import parallel
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    from multimodal_vae_comparison_amd.synthetic import workload
    desc, cfg, dims, data, meta = workload(name, batch, device=dev, seed=1 + rank)
    torch.manual_seed(0)                                   # identical replicas on every rank
    tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
    tr.model.train()
    tr.configure_optimizers()
    if path_world > 1:
        # broadcast of the flat parameters, 1/world folded into Adam, per-rank noise / dropout streams
        parallel.setup_replica(tr, rank, world)
    tr.capture(data, path_world, input_ring=input_ring)       # world 1: the Adam step is part of the captured graph
    return tr, desc, meta


def _self_launch(a):
    """`python bench.py --gpus N` as a BARE command (no torchrun around it, WORLD_SIZE unset): start the N ranks ourselves,
    exactly as the driver would -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>` -- as a CHILD process, launched before this process has made any GPU
    call (never a re-exec: a process that has initialised the GPU must not be replaced), relay rank 0's one JSON line and
    the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    for l in r.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    sys.exit(r.returncode if r.returncode or lines else 1)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(a)
    from multimodal_vae_comparison_amd import parallel
    from multimodal_vae_comparison_amd.synthetic import step_flops_per_sample, workload
    # torchrun: the process group comes up before anything touches the GPU
    ndev = torch.cuda.device_count()           # (counting devices does not initialise the GPU)
    local_env = int(os.environ.get("LOCAL_RANK", 0))
    dev_index = local_env % max(ndev, 1) if a.backend == "gloo" else local_env
    rank, local, world = parallel.init_from_env(a.backend, torch.device("cuda", dev_index))
    assert world == a.gpus or world == 1 and a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # --force-collective (testing on one GPU): take the multi-GPU structure of the step -- graph without the optimiser,
    # one RCCL all-reduce of the flat gradients, separate Adam launch -- with a single-rank process group
    path_world = world
    if a.force_collective and world == 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1, device_id=dev)
        path_world = 2

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    tr, desc, meta = _build(a.config, a.batch, dev, rank, world, path_world)
    B = meta["B"]
    # Set-up, untimed and reported (`config.settle_steps`): the first ~25 replays of a freshly instantiated graph on a fresh
    # box run ~2 % slow (clocks, instruction / constant caches, first-touch of the arena): with the driver's `--steps 20
    # --warmup 5` the timed region would otherwise BE that transient (measured: 0.396 vs 0.388 ms on the same box, the three
    # `spread` repeats right behind it 0.388).  The W warm-up steps and the K timed steps follow unchanged.
    # (round 6) the caller's protocol on the fresh graph is measured FIRST and reported beside the headline
    # (`first_window_ms_per_step`: W warm-ups, K timed steps, nothing in front), so that lines of different rounds can be
    # compared on either footing; `untimed_steps_before_timed_region` is everything that ran before the headline's K steps.
    settle = max(0, a.settle)
    first_dt = None
    if settle > 0:
        first_dt, _ = _timed(tr, a.steps, a.warmup, path_world, barrier)
        for _ in range(settle):
            tr.fused_step(path_world)
        barrier()
    dt, out = _timed(tr, a.steps, a.warmup, path_world, barrier)
    if first_dt is None:
        first_dt = dt
    if world > 1:
        t = torch.tensor([first_dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        first_dt = float(t.item())
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss"].item())
    in_sync = None
    if world > 1:       # replicas must still hold identical parameters: compare a checksum across the ranks
        cs = tr.flat.data.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        in_sync = bool(lo.item() == hi.item())

    res = None
    if rank == 0:
        flops = step_flops_per_sample(meta)
        sps = a.steps * B * world / dt
        res = {"metric": METRIC.get(a.config, f"training samples/sec, {a.config} (fwd+bwd+Adam)"), "value": round(sps, 1),
               "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * dt / a.steps, 4),
               "first_window_ms_per_step": round(1e3 * first_dt / a.steps, 4),
               "untimed_steps_before_timed_region": (a.warmup + a.steps + settle if settle > 0 else 0) + a.warmup,
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{desc}; Adam(amsgrad) lr 1e-4, beta 1, train mode (dropout on)",
                          "global_batch": B * world, "parallelism": f"dp{world}", "settle_steps": settle,
                          "abi_calls_per_step": getattr(tr, "abi_calls_in_graph", None),
                          "step_mflop_per_sample": round(flops / 1e6, 1),
                          "step_flops_fraction_of_f32_mfma_peak": round(
                              sps / world * flops / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
                          # dtype f32 = the arithmetic contract of every kernel: fp32 operands, fp32 accumulation, the
                          # reference's results to 1e-4.  The gather-form convolutions meet it with split-bf16 MFMAs
                          "arithmetic": "fp32 operands / accumulate / results everywhere; the gather-form convolutions "
                                        "form each fp32 product from six bf16 MFMAs on three exact bf16 terms per operand "
                                        "(error vs fp64 as the fp32-MFMA kernels, DESIGN 5d)"},
               "final_loss": round(loss, 3)}
        if world > 1 or path_world > 1:
            # top level, so that the driver's scaling record can read them: which collective path the step took and
            # whether the replicas still hold bit-identical parameters after the timed steps
            in_graph = bool(getattr(tr, "_collective_in_graph", False))
            res["collective"] = {"path": "in graph" if in_graph else "after graph", "backend": a.backend,
                                 "what": ("one all-reduce of the flat gradient buffer per step, captured into the step's "
                                          "hipGraph with the Adam launch (validated by one replay + cross-rank checksum "
                                          "at capture)" if in_graph else
                                          "one all-reduce of the flat gradient buffer per step after the graph, then the "
                                          "fused Adam launch"),
                                 "bytes": int(tr.flat.grad.numel() * 4),
                                 # ranks of the communicator the all-reduce runs on, as the process group reports it
                                 "nranks": (torch.distributed.get_world_size() if torch.distributed.is_initialized()
                                            else 1)}
            stager = getattr(tr, "_stager", None)
            if in_graph and stager is not None:
                res["collective"]["what"] = (f"{stager.n_collectives} bucketed all-reduces of the flat gradient buffer per step, "
                                             "sent while the ResNet tower's backward pass is still running "
                                             "(parallel.StagedGradReducer, >= 25 MB buckets), captured into the step's "
                                             "hipGraph with the Adam launch")
            res["collective_bytes"] = res["collective"]["bytes"]
            res["n_collectives"] = stager.n_collectives if (in_graph and stager is not None) else 1
            res["replicas_in_sync"] = in_sync
            res["config"]["collective"] = res["collective"]["what"]
    if world == 1 and rank == 0:
        # the timed region above is short (K x ~0.4 ms); three more timed repeats of the same K steps say how much one
        # reading moves on THIS box (box-to-box spread is larger: profiles/ keeps the builder boxes' lines)
        reps = [round(1e3 * _timed(tr, a.steps, 0, path_world, barrier)[0] / a.steps, 4) for _ in range(3)]
        res["spread"] = {"ms_per_step_repeats": reps, "steps": a.steps,
                         "what": "three further timed repeats of the same K steps on this box (not the headline value)"}
        res["roofline"] = dominant_kernel_roofline(meta, dev)
        # context for the dominant-kernel figure: the WHOLE step's algorithmic FLOPs over its wall time, same peak
        res["roofline"]["step_frac"] = res["config"]["step_flops_fraction_of_f32_mfma_peak"]
        res["roofline"]["step_achieved"] = round(sps * flops / 1e12, 2)
        if not a.no_extras and path_world == 1 and a.config == "cfg2":
            res["extras"] = extras(tr, a, dev, barrier)
        if not a.no_cpu_baseline:
            _, _, _, batch_cpu, _ = workload(a.config, a.batch, device="cpu", seed=1)
            res["cpu_baseline"] = cpu_baseline(meta, batch_cpu)
            res["config"]["gpu_over_cpu"] = round(sps / res["cpu_baseline"]["value"], 1)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


def extras(tr, a, dev, barrier):
    """non-headline figures (never `value`): (1) the same step with the input step in the loop -- a fresh batch in the
    compact host format (uint8 pixels, int32 token ids) copied H2D from pinned memory and expanded on the device
    (MultimodalVAE.load_batch_compact) before every replay; (2) large per-GPU batches (512 and 1000 = the reference's
    positional-table cap, nn_modules.py:419)"""
    from multimodal_vae_comparison_amd.synthetic import step_flops_per_sample
    out = {}
    B, T, V = 128, 32, 27
    g = torch.Generator().manual_seed(3)
    host = []
    for _ in range(4):          # a small ring of pinned host batches
        u8 = torch.randint(0, 256, (B, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory()
        tok = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32).pin_memory()
        lens = torch.randint(3, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[0] = T
        host.append({"mod_1": {"u8": u8}, "mod_2": {"tokens": tok, "lengths": lens.pin_memory()}})
    it = {"i": 0}

    def feed():
        tr.load_batch_compact(host[it["i"] & 3])
        it["i"] += 1

    steps = max(20, a.steps // 2)
    dt, _ = _timed(tr, steps, 25, 1, barrier, pre=feed)     # (25 warm-up steps: stream / staging-buffer set-up is one-time)
    out["with_input_pipeline_serial"] = {"value": round(steps * B / dt, 1), "unit": "samples/s",
                                         "ms_per_step": round(1e3 * dt / steps, 4),
                                         "what": "pinned uint8 image + int32 token batch (1.6 MB) H2D, device expansion, "
                                                 "captured step, one after the other on one stream, every step"}
    host = [tr.pack_compact_pinned(h) for h in host]      # one pinned buffer, one H2D copy per batch
    pipe = tr.input_pipe(host[0])

    def feed_pipe():                # expand batch i (copied under step i-1), start copying batch i+1: one library call
        it["i"] += 1
        pipe.step(host[it["i"] & 3])
    pipe.prefetch(host[0])
    dt, _ = _timed(tr, steps, 25, 1, barrier, pre=feed_pipe)
    pipe.close()
    out["with_input_pipeline"] = {"value": round(steps * B / dt, 1), "unit": "samples/s",
                                  "ms_per_step": round(1e3 * dt / steps, 4),
                                  "what": "a fresh compact batch every step: ONE H2D copy of batch i+1 (packed pinned "
                                          "buffer) on a copy stream under step i, device expansion + captured step on "
                                          "the main stream; the input step is one native call (csrc/input_pipe.hip, "
                                          "MultimodalVAE.input_pipe)"}
    # the input step INSIDE the graph: head = expansion of the staged batch, tail = pull of the next ring slot on the text
    # tower's stream; no runtime call per step besides the graph launch (MultimodalVAE.capture(..., input_ring=...))
    t3, _, _ = _build("cfg2", B, dev, 0, 1, 1, input_ring=host)
    dt, _ = _timed(t3, steps, 25, 1, barrier)
    out["with_input_pipeline_in_graph"] = {"value": round(steps * B / dt, 1), "unit": "samples/s",
                                           "ms_per_step": round(1e3 * dt / steps, 4),
                                           "what": "a fresh compact batch every step, the input step captured into the "
                                                   "step's hipGraph: expansion launches at its head, the pull of the next "
                                                   "pinned ring slot over the host link at the tail of the text tower's "
                                                   "stream, slot counter on the device (GraphInputRing)"}
    del t3
    lb = {}
    for Bl in (512, 1000):
        t2, _, meta = _build("cfg2", Bl, dev, 0, 1, 1)
        dt, _ = _timed(t2, 30, 5, 1, barrier)
        sps = 30 * Bl / dt
        lb[str(Bl)] = {"value": round(sps, 1), "unit": "samples/s", "ms_per_step": round(1e3 * dt / 30, 4),
                       "step_flops_fraction_of_f32_mfma_peak": round(
                           sps * step_flops_per_sample(meta) / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)}
        del t2
    out["large_batch"] = lb
    return out


if __name__ == "__main__":
    main()
