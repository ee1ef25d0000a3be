"""GPU parity of every C-ABI kernel family against a plain PyTorch fp32 reference of the same op
(computed on the CPU in float64 where cheap).  Tolerance: 1e-4 relative to the tensor's max magnitude
(north-star: 1e-4 relative fp32), tighter where the op is elementwise."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def check(a, b, tol, what):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert math.isfinite(e) and e <= tol, f"{what}: rel err {e:.3e} > {tol}"


def _act(x, act):
    return {0: lambda t: t, 1: F.silu, 2: torch.relu, 3: F.gelu}[act](x)


@pytest.fixture(scope="module")
def ops(hip_lib):
    from multimodal_vae_comparison_amd import ops
    return ops


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,Hin,act", [(3, 3, 64, 0), (5, 32, 32, 1), (6, 32, 16, 1), (5, 32, 8, 1), (128, 32, 8, 2),
                                           (2, 32, 32, 0), (130, 32, 16, 1), (64, 3, 64, 0),
                                           # shapes served by the split-bf16 kernels (conv_gather_b16.inc): every geometry
                                           (70, 32, 32, 2), (70, 32, 32, 1), (520, 32, 16, 2), (33, 3, 64, 0),
                                           # 2080 position tiles: the fused backward with the split-bf16 SCATTER body as its
                                           # data-gradient half (conv2d_bwd_fused_b16_kernel, from 2048 tiles on)
                                           (260, 32, 32, 1)])
def test_conv2d_fwd_bwd(ops, B, Cin, Hin, act):
    g = torch.Generator().manual_seed(B * 1000 + Hin)
    x = torch.randn(B, Cin, Hin, Hin, generator=g)
    w = torch.randn(32, Cin, 4, 4, generator=g) * (1.0 / math.sqrt(Cin * 16))
    b = torch.randn(32, generator=g) * 0.1
    dy = torch.randn(B, 32, Hin // 2, Hin // 2, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(_act(xr, act), wr, br, stride=2, padding=1)
    yr.backward(dy.double())
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d_k4s2(xg, wg, bg, act)
    y.backward(dy.to(DEV))
    check(y, yr, 2e-5, "y")
    check(wg.grad, wr.grad, 5e-5, "dw")
    check(bg.grad, br.grad, 5e-5, "db")
    check(xg.grad, xr.grad, 5e-5, "dx")
    # accumulate-into-preset-gradient path
    gw, gb = torch.ones_like(wg), torch.ones_like(bg)
    x2 = x.to(DEV).requires_grad_(True)
    ops.conv2d_k4s2(x2, wg.detach(), bg.detach(), act, gw, gb).backward(dy.to(DEV))
    check(gw - 1, wr.grad, 5e-5, "dw (accumulated)")
    check(gb - 1, br.grad, 5e-5, "db (accumulated)")


@pytest.mark.parametrize("B,Cout,Hin,act,ep", [(3, 32, 4, 2, 0), (5, 32, 8, 2, 0), (6, 32, 16, 2, 0), (128, 32, 4, 0, 0),
                                               (4, 3, 32, 2, 6), (130, 32, 8, 2, 0), (33, 3, 32, 2, 6), (7, 3, 16, 0, 0),
                                               # split-bf16: the fused backward's data-gradient body on every geometry,
                                               # the 32 -> 3 forward (conv_scatter3_b16.inc)
                                               (70, 32, 16, 2, 0), (520, 32, 8, 2, 0), (70, 3, 32, 2, 6), (66, 3, 32, 0, 0)])
def test_convT2d_fwd_bwd(ops, B, Cout, Hin, act, ep):
    g = torch.Generator().manual_seed(B * 1000 + Hin + Cout)
    x = torch.randn(B, 32, Hin, Hin, generator=g)
    w = torch.randn(32, Cout, 4, 4, generator=g) * (1.0 / math.sqrt(32 * 4))
    b = torch.randn(Cout, generator=g) * 0.1
    dy = torch.randn(B, Cout, 2 * Hin, 2 * Hin, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose2d(_act(xr, act), wr, br, stride=2, padding=1)
    if ep == 6:
        yr = torch.sigmoid(yr).clamp(1e-6, 1 - 1e-6)
    yr.backward(dy.double())
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.convT2d_k4s2(xg, wg, bg, act, ep)
    y.backward(dy.to(DEV))
    check(y, yr, 2e-5, "y")
    check(wg.grad, wr.grad, 5e-5, "dw")
    check(bg.grad, br.grad, 5e-5, "db")
    check(xg.grad, xr.grad, 5e-5, "dx")


@pytest.mark.parametrize("B,act", [(4, 2), (33, 2), (128, 2), (300, 2), (520, 2), (5, 0), (6, 1)])
def test_convT3_bce_fused_matches_float64(ops, B, act):
    """Dec_CNN's last layer + clamp(sigmoid) + bce row sums + d loss / d logits in ONE launch (csrc/conv_t3.inc): rows, and
    the layer's dx / dw / db through the stored logit gradient, against float64 torch -- every strip plan (4 / 2 strips per
    image with the ticketed row sum, whole images), announced seed and another upstream gradient."""
    g = torch.Generator().manual_seed(B + 17)
    x = torch.randn(B, 32, 32, 32, generator=g)
    w = torch.randn(32, 3, 4, 4, generator=g) * (1.0 / math.sqrt(32 * 4))
    b = torch.randn(3, generator=g) * 0.1
    tgt = torch.rand(B, 3, 64, 64, generator=g)
    tgt[:, :, :8] = (tgt[:, :, :8] > 0.5).float()          # exact 0 / 1 targets too (images are mostly saturated)
    x[0, :, 3, 5] *= 40.0                                  # logits far out: the clamp is active there
    seed = 0.37
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    xh = torch.sigmoid(F.conv_transpose2d(_act(xr, act), wr, br, stride=2, padding=1)).clamp(1e-6, 1 - 1e-6)
    rows_r = F.binary_cross_entropy(xh, tgt.double(), reduction="none").sum((1, 2, 3))
    for other in (False, True):
        up = torch.full((B,), seed) if not other else torch.linspace(0.1, 1.0, B)
        for t in (xr, wr, br):
            t.grad = None
        rows_r.backward(up.double(), retain_graph=True)
        xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        seed_t = torch.full((B,), seed, device=DEV)
        with ops.ConstSeed(seed_t, seed):
            assert ops.convT3_bce_supported(xg, wg, tgt.to(DEV)) == (B >= 256)      # (the dispatch rule; the op takes any batch)
            rows = ops.convT3_bce(xg, wg, bg, act, None, None, tgt.to(DEV))
        rows.backward(seed_t if not other else up.to(DEV))
        check(rows, rows_r, 1e-5, "bce rows")
        check(xg.grad, xr.grad, 5e-5, "dx")
        check(wg.grad, wr.grad, 5e-5, "dw")
        check(bg.grad, br.grad, 5e-5, "db")


# ---------------------------------------------------------------------------------------------
# split-bf16 == fp32 (VERDICT r4 "What's weak" #2): every shape the split-bf16 bodies serve, held to the accuracy that only a
# THREE-term split reaches -- 2e-6 of the tensor maximum against fp64 (measured 2-6e-7; a two-term split sits at ~1e-5) -- and
# to 1e-5 of the fp32-MFMA kernel on the same inputs (mmvae_conv_plan(0) selects that core in the same process).
SPLIT_BF16_CONV = [(70, 32, 32, 2), (70, 32, 32, 1), (520, 32, 16, 2), (33, 3, 64, 0), (260, 32, 32, 1), (130, 32, 16, 1)]
SPLIT_BF16_CONVT = [(70, 32, 16, 2, 0), (520, 32, 8, 2, 0), (70, 3, 32, 2, 6), (66, 3, 32, 0, 0), (130, 32, 8, 2, 0)]


def _conv_case(ops, kind, B, C, Hin, act, ep, scale=1.0, poke=None):
    """(y, dx) of one layer on the GPU + the fp64 results; kind 'conv' (C = Cin -> 32) or 'convT' (32 -> C = Cout)"""
    g = torch.Generator().manual_seed(B * 1000 + Hin + C)
    if kind == "conv":
        x = torch.randn(B, C, Hin, Hin, generator=g)
        w = torch.randn(32, C, 4, 4, generator=g) * (1.0 / math.sqrt(C * 16))
        b = torch.randn(32, generator=g) * 0.1
        dy = torch.randn(B, 32, Hin // 2, Hin // 2, generator=g)
    else:
        x = torch.randn(B, 32, Hin, Hin, generator=g)
        w = torch.randn(32, C, 4, 4, generator=g) * (1.0 / math.sqrt(32 * 4))
        b = torch.randn(C, generator=g) * 0.1
        dy = torch.randn(B, C, 2 * Hin, 2 * Hin, generator=g)
    x, b, dy = x * scale, b * scale, dy * scale
    if poke is not None:
        poke(x)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    if kind == "conv":
        yr = F.conv2d(_act(xr, act), wr, br, stride=2, padding=1)
    else:
        yr = F.conv_transpose2d(_act(xr, act), wr, br, stride=2, padding=1)
        if ep == 6:
            yr = torch.sigmoid(yr).clamp(1e-6, 1 - 1e-6)
    yr.backward(dy.double())

    def run():
        xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d_k4s2(xg, wg, bg, act) if kind == "conv" else ops.convT2d_k4s2(xg, wg, bg, act, ep)
        y.backward(dy.to(DEV))
        return y.detach(), xg.grad.detach()
    return run, yr.detach(), xr.grad.detach()


def _both_cores(hip_lib, run):
    old = hip_lib.mmvae_conv_plan(1)
    try:
        y16, dx16 = run()
        hip_lib.mmvae_conv_plan(0)
        y32, dx32 = run()
    finally:
        hip_lib.mmvae_conv_plan(old)
    return y16, dx16, y32, dx32


@pytest.mark.parametrize("kind,B,C,Hin,act,ep", [("conv",) + c + (0,) for c in SPLIT_BF16_CONV] +
                         [("convT",) + c for c in SPLIT_BF16_CONVT])
def test_split_bf16_is_fp32_equivalent(ops, hip_lib, kind, B, C, Hin, act, ep):
    run, yr, dxr = _conv_case(ops, kind, B, C, Hin, act, ep)
    y16, dx16, y32, dx32 = _both_cores(hip_lib, run)
    check(y16, yr, 2e-6, "split-bf16 y vs fp64")
    check(dx16, dxr, 2e-6, "split-bf16 dx vs fp64")
    check(y32, yr, 2e-6, "fp32-MFMA y vs fp64")
    check(dx32, dxr, 2e-6, "fp32-MFMA dx vs fp64")
    check(y16, y32, 1e-5, "split-bf16 y vs the fp32-MFMA kernel")
    check(dx16, dx32, 1e-5, "split-bf16 dx vs the fp32-MFMA kernel")


@pytest.mark.parametrize("kind,B,C,Hin,act,ep", [("conv", 70, 32, 32, 0, 0), ("convT", 70, 32, 16, 0, 0)])
def test_split_bf16_tiny_magnitudes(ops, hip_lib, kind, B, C, Hin, act, ep):
    """Operands of magnitude ~1e-30 (far below anything a training step holds, far above 2^-110 where the third term of the
    split becomes subnormal): full accuracy.  Operands at the bottom of the normal range (~1e-37): the lower two terms of
    the split are subnormal and may flush, the documented floor is 2^-8 relative (split-bf16 and non-finite
    values); both cores must stay finite."""
    run, yr, dxr = _conv_case(ops, kind, B, C, Hin, act, ep, scale=1e-30)
    y16, dx16, _, _ = _both_cores(hip_lib, run)
    check(y16, yr, 2e-6, "split-bf16 y at 1e-30")
    check(dx16, dxr, 2e-6, "split-bf16 dx at 1e-30")
    run, yr, dxr = _conv_case(ops, kind, B, C, Hin, act, ep, scale=2e-37)
    y16, dx16, y32, dx32 = _both_cores(hip_lib, run)
    for t in (y16, dx16, y32, dx32):
        assert bool(torch.isfinite(t).all())
    check(y16, yr, 2.0 ** -8, "split-bf16 y at 2e-37")
    check(dx16, dxr, 2.0 ** -8, "split-bf16 dx at 2e-37")


@pytest.mark.parametrize("kind,B,C,Hin", [("conv", 70, 32, 32), ("convT", 70, 32, 16)])
def test_split_bf16_non_finite_inputs(ops, hip_lib, kind, B, C, Hin):
    """One +Inf and one -Inf input element.  The chosen behaviour: an output the element reaches is NON-FINITE on
    both cores (the fp32-MFMA kernel gives +-Inf or NaN as IEEE does; the split-bf16 kernel gives NaN, since Inf - bf16(Inf)
    is NaN) -- a non-finite activation is never turned into a finite one -- and every output it does not reach is untouched."""
    def poke(x):
        x[3, 5 % x.shape[1], 7, 9] = float("inf")
        x[11, 1, 2, 3] = float("-inf")
    run, yr, _ = _conv_case(ops, kind, B, C, Hin, 0, 0, poke=poke)
    y16, _, y32, _ = _both_cores(hip_lib, run)
    reach = ~torch.isfinite(yr)                       # fp64: exactly the outputs the two elements reach
    assert int(reach.sum()) > 0
    for name, y in (("split-bf16", y16), ("fp32-MFMA", y32)):
        y = y.cpu()
        assert not bool(torch.isfinite(y[reach]).any()), name
        assert bool(torch.isfinite(y[~reach]).all()), name
        e = float((y[~reach].double() - yr[~reach]).abs().max() / float(yr[~reach].abs().max()))
        assert e <= 2e-6, (name, e)


@pytest.mark.parametrize("M,K,N,act", [(128, 512, 512, 1), (6, 8, 512, 0), (7, 512, 64, 2), (4096, 54, 162, 0),
                                       (160, 128, 54, 3), (33000, 54, 128, 0), (5, 54, 16, 0), (300, 32, 27, 0),
                                       # the large-tile kernel (gemm_big_kernel): MNIST towers at K*B rows, ResNet shapes,
                                       # ragged edges in M / N / K, every orientation through fwd / dgrad / wgrad
                                       (7680, 400, 400, 2), (7680, 400, 784, 2), (7680, 20, 400, 0), (6144, 576, 64, 2),
                                       (1000, 2048, 512, 2), (32000, 56, 160, 0), (520, 1024, 256, 3), (3844, 36, 132, 1),
                                       # 64 x 64 tiles (round 3): the 512-wide image-tower linears at batch 512 .. 1000
                                       (1000, 512, 512, 2), (512, 512, 512, 0), (1000, 32, 512, 0), (996, 512, 64, 1), (300, 516, 100, 1),
                                       # one wave per 32 rows (round 6): the action towers' 32 -> 96 / 32 projections at 12 800 rows
                                       (12800, 32, 96, 0), (1030, 32, 32, 0), (3200, 32, 128, 0), (2049, 32, 64, 0)])
def test_linear_fwd_bwd(ops, M, K, N, act):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(_act(xr, act), wr, br)
    yr.backward(dy.double())
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xg, wg, bg, act)
    y.backward(dy.to(DEV))
    check(y, yr, 2e-5, "y")
    check(wg.grad, wr.grad, 5e-5, "dw")
    check(bg.grad, br.grad, 5e-5, "db")
    check(xg.grad, xr.grad, 5e-5, "dx")


@pytest.mark.parametrize("M,K,N,act", [(1, 4, 1, 0), (17, 20, 33, 1), (128, 1024, 64, 0), (256, 36, 512, 2), (130, 516, 100, 1),
                                       (64, 54, 54, 0), (3, 1028, 5, 0), (200, 8, 8, 3), (16, 512, 2048, 1)])
def test_linear_small_batch_paths(ops, M, K, N, act):
    """shapes around the dispatch boundaries of the register-operand GEMMs (16x16 / 32x32 tiles, float4 vs strided
    operand loads for K % 4 != 0, reductions longer than 1024 back on the staged kernel) and the fused backward"""
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    F.linear(_act(xr, act), wr, br).backward(dy.double())
    for preset in (False, True):      # gradients returned to autograd / accumulated into preset views
        xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        gw = torch.ones(N, K, device=DEV) if preset else None
        gb = torch.ones(N, device=DEV) if preset else None
        y = ops.linear(xg, wg, bg, act, gw, gb)
        y.backward(dy.to(DEV))
        check(y, F.linear(_act(x.double(), act), w.double(), b.double()), 2e-5, "y")
        check(xg.grad, xr.grad, 5e-5, "dx")
        check(gw - 1 if preset else wg.grad, wr.grad, 5e-5, "dw")
        check(gb - 1 if preset else bg.grad, br.grad, 5e-5, "db")


@pytest.mark.parametrize("M,K,N,act", [(2048, 512, 512, 1), (3000, 512, 512, 2), (2100, 784, 400, 0), (2052, 132, 260, 3),
                                       (4096, 128, 128, 0), (2049, 516, 1028, 2)])
def test_linear_tiled_split_bf16_is_fp32_equivalent(ops, hip_lib, M, K, N, act):
    """csrc/gemm_b16.inc: the LDS-tiled split-bf16 kernels that take wide Linear layers (N, K >= 128) at 2048 .. 4096 rows
    -- forward, data gradient, and the grouped data + weight gradient launch with its row-split partials -- against fp64 at
    3e-6 of each tensor's maximum (a two-term split would sit at ~1e-5), against the fp32-MFMA bodies on the same inputs,
    ragged edges in every dimension, gradients returned to autograd and accumulated into preset views"""
    g = torch.Generator().manual_seed(M + 3 * K + 7 * N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(_act(xr, act), wr, br)
    yr.backward(dy.double())
    outs = {}
    was = hip_lib.mmvae_gemm_b16_set(1)
    try:
        for core in (1, 0):
            hip_lib.mmvae_gemm_b16_set(core)
            assert hip_lib.mmvae_linear_bwd_splits(M, N, K) >= 1
            for preset in (False, True):
                xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
                gw = torch.ones(N, K, device=DEV) if preset else None
                gb = torch.ones(N, device=DEV) if preset else None
                y = ops.linear(xg, wg, bg, act, gw, gb)
                y.backward(dy.to(DEV))
                torch.cuda.synchronize()
                outs[(core, preset)] = (y.detach(), xg.grad, gw - 1 if preset else wg.grad, gb - 1 if preset else bg.grad)
    finally:
        hip_lib.mmvae_gemm_b16_set(was)
    refs = (yr.detach(), xr.grad, wr.grad, br.grad)
    for (core, preset), got in outs.items():
        tol = 3e-6 if core else 5e-5
        for name, a_, r_ in zip(("y", "dx", "dw", "db"), got, refs):
            check(a_, r_, tol if name != "db" else 5e-6 if core else 5e-5, f"{name} (b16={core}, preset={preset})")
    for name, a_, b_ in zip(("y", "dx", "dw", "db"), outs[(1, True)], outs[(0, True)]):
        check(a_, b_.double(), 2e-5, f"{name}: split-bf16 vs fp32-MFMA")


@pytest.mark.parametrize("M,p", [(12800, 0.1), (100, 0.0), (33, 0.3)])
def test_proj32_layernorm_fused_matches_two_launches(ops, monkeypatch, M, p):
    """mmvae_proj32_ln_fwd: LayerNorm(dropout(x W^T + b) + r) for d_model 32 in ONE launch (the Linear's output is a
    placeholder that only the LayerNorm consumes) against the two launches it replaces -- same dropout mask, saved tensors
    and backward -- and against torch in fp64 with the extracted mask; ragged row counts"""
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, 32, generator=g).to(DEV)
    r = torch.randn(M, 32, generator=g).to(DEV)
    w = (torch.randn(32, 32, generator=g) * 0.3).to(DEV)
    b = torch.randn(32, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(32, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(32, generator=g)).to(DEV)
    dy = torch.randn(M, 32, generator=g).to(DEV)
    drop, mask = None, torch.ones(M, 32, dtype=torch.float64, device=DEV)
    if p > 0:
        st = DropoutState().to(DEV)
        slot, call = st.begin()
        drop = st.spec(slot, call, 2, p, "drop1")
        mask = ops.dropout_mask(drop, M * 32).double().view(M, 32)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "PROJ32_LN", fused)
        xs, rs, ws, bs, gs, bts = (t.clone().requires_grad_(True) for t in (x, r, w, b, gamma, beta))
        calls0 = ops.CALLS[0]
        y = ops.layernorm_residual(ops.linear(xs, ws, bs, lazy=fused), rs, gs, bts, None, None, drop)
        n_fwd = ops.CALLS[0] - calls0
        y.backward(dy)
        outs.append((y.detach(), xs.grad, rs.grad, ws.grad, bs.grad, gs.grad, bts.grad, n_fwd))
    assert outs[0][7] == 1 and outs[1][7] == 2
    xr, rr, wr, br, gr, btr = (t.double().requires_grad_(True) for t in (x, r, w, b, gamma, beta))
    ref = F.layer_norm((xr @ wr.t() + br) * mask + rr, (32,), gr, btr, 1e-5)
    ref.backward(dy.double())
    names = ("y", "dx", "dr", "dW", "db", "dgamma", "dbeta")
    for k, name in enumerate(names):
        check(outs[0][k], outs[1][k].double(), 3e-6, f"{name}: fused vs two launches")
    for k, (name, t) in enumerate(zip(names, (ref.detach(), xr.grad, rr.grad, wr.grad, br.grad, gr.grad, btr.grad))):
        check(outs[0][k], t, 2e-5, f"{name} vs fp64")


@pytest.mark.parametrize("M,FF,p", [(12800, 1024, 0.1), (70, 128, 0.0), (33, 96, 0.3)])
def test_ffn32_layernorm_epilogue_matches_two_launches(ops, monkeypatch, M, FF, p):
    """mmvae_ffn32_fwd_b16_ln: LayerNorm(dropout(ffn(x)) + x) with the LayerNorm as the feed-forward launch's epilogue
    against the two launches (same masks, same saved tensors, same backward): outputs and every gradient"""
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    g = torch.Generator().manual_seed(M + FF)
    x = torch.randn(M, 32, generator=g).to(DEV)
    w1 = (torch.randn(FF, 32, generator=g) * 0.3).to(DEV)
    b1 = (torch.randn(FF, generator=g) * 0.3).to(DEV)
    w2 = (torch.randn(32, FF, generator=g) * 0.1).to(DEV)
    b2 = torch.randn(32, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(32, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(32, generator=g)).to(DEV)
    dy = torch.randn(M, 32, generator=g).to(DEV)
    d_ffn = d_ln = None
    if p > 0:
        st = DropoutState().to(DEV)
        slot, call = st.begin()
        d_ffn, d_ln = st.spec(slot, call, 3, p, "ffn"), st.spec(slot, call, 4, p, "drop2")
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "FFN32_LN", fused)
        xs, w1s, b1s, w2s, b2s, gs, bts = (t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gamma, beta))
        calls0 = ops.CALLS[0]
        h = ops.ffn32(xs, w1s, b1s, w2s, b2s, d_ffn, None, None, (xs, gs, bts, d_ln))
        y = ops.layernorm_residual(h, xs, gs, bts, None, None, d_ln)
        n_fwd = ops.CALLS[0] - calls0
        y.backward(dy)
        outs.append((y.detach(), xs.grad, w1s.grad, b1s.grad, w2s.grad, b2s.grad, gs.grad, bts.grad, n_fwd))
    assert outs[1][8] == outs[0][8] + 1, (outs[0][8], outs[1][8])
    for k, name in enumerate(("y", "dx", "dW1", "db1", "dW2", "db2", "dgamma", "dbeta")):
        check(outs[0][k], outs[1][k].double(), 3e-6, f"{name}: epilogue vs two launches")


def test_input_expansion_bit_exact(ops):
    """uint8 pixels / 255 and token ids -> one-hot + mask on the device, bit-identical to the
    reference's host preprocessing (torch.tensor(uint8) / 255; one_hot_encode + lengths_to_mask)"""
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (7, 3, 64, 64), generator=g, dtype=torch.uint8)
    u8.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)          # every byte value
    ref = u8 / 255
    out = ops.expand_image_u8(u8.to(DEV), torch.empty(7, 3, 64, 64, device=DEV))
    assert torch.equal(out.cpu(), ref)
    odd = u8.view(-1)[1:1 + 1001].clone()                             # unaligned source, length not a multiple of 4
    out = ops.expand_image_u8(odd.to(DEV), torch.empty(1001, device=DEV))
    assert torch.equal(out.cpu(), odd / 255)
    B, T, V = 9, 13, 27
    tok = torch.randint(-1, V, (B, T), generator=g, dtype=torch.int32)
    lens = torch.randint(0, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    oh = torch.full((B, T, V), 7.0, device=DEV)
    mask = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    ops.expand_text_tokens(tok.to(DEV), lens.to(DEV), oh, mask.view(torch.uint8))
    mref = torch.arange(T)[None, :] < lens[:, None]
    ohref = torch.zeros(B, T, V)
    for b in range(B):
        for t in range(int(lens[b])):
            if tok[b, t] >= 0:
                ohref[b, t, tok[b, t]] = 1.0
    assert torch.equal(oh.cpu(), ohref) and torch.equal(mask.cpu(), mref)


def test_head_softmax(ops):
    g = torch.Generator().manual_seed(1)
    h = torch.randn(37, 2 * 42, generator=g)
    dh = torch.randn(37, 84, generator=g)
    hr = h.double().requires_grad_(True)
    outr = torch.cat([hr[:, :42], F.softmax(hr[:, 42:], -1) + 1e-6], -1)
    outr.backward(dh.double())
    hg = h.to(DEV).requires_grad_(True)
    out = ops.head_softmax(hg * 1.0)
    out.backward(dh.to(DEV))
    check(out, outr, 1e-6, "out")
    check(hg.grad, hr.grad, 1e-5, "dh")


def _poe_ref(theta, packed, eps, with_prior, kl_mask):
    D = theta.shape[1]
    sp = F.softmax(theta, dim=1) * D
    mus = [p[:, :D] for p in packed]
    lvs = [p[:, D:] for p in packed]
    T = [1.0 / (torch.exp(l) + 1e-8) for l in lvs]
    P = sum(T) + (1.0 / (1.0 + 1e-8) if with_prior else 0.0)
    muJ = sum(m * t for m, t in zip(mus, T)) / P
    varJ = 1.0 / P

    def kl(mu, s):
        vr = (s / sp) ** 2
        return (0.5 * (vr + (mu / sp) ** 2 - 1 - vr.log())).sum(-1)

    E = len(packed)
    rows = [kl(m, l) if kl_mask >> e & 1 else torch.zeros(m.shape[0], dtype=m.dtype) for e, (m, l) in enumerate(zip(mus, lvs))]
    rows.append(kl(muJ, varJ) if kl_mask >> E & 1 else torch.zeros(muJ.shape[0], dtype=muJ.dtype))
    z = torch.stack([muJ + varJ * e for e in eps]) if eps else torch.zeros(0, *muJ.shape)
    return torch.stack([muJ, varJ]), torch.stack(rows), z


@pytest.mark.parametrize("E,n_z,with_prior,kl_mask,B,D", [(2, 2, True, 0b111, 128, 32), (1, 1, True, 0b10, 7, 8),
                                                          (3, 3, True, 0b1111, 130, 42), (2, 0, False, 0, 5, 16),
                                                          (2, 1, True, 0b100, 300, 70)])
def test_poe_reparam_kl(ops, E, n_z, with_prior, kl_mask, B, D):
    g = torch.Generator().manual_seed(E * 100 + B)
    packed = [torch.cat([torch.randn(B, D, generator=g), F.softmax(torch.randn(B, D, generator=g), -1) + 1e-6], -1)
              for _ in range(E)]
    eps = [torch.randn(B, D, generator=g) for _ in range(n_z)]
    theta = torch.randn(1, D, generator=g) * 0.3
    gkl = torch.randn(E + 1, B, generator=g)
    gz = torch.randn(n_z, B, D, generator=g)
    pr = [p.double().requires_grad_(True) for p in packed]
    tr = theta.double().requires_grad_(True)
    jr, klr, zr = _poe_ref(tr, pr, [e.double() for e in eps], with_prior, kl_mask)
    tot = (klr * gkl.double()).sum() + ((zr * gz.double()).sum() if n_z else 0.0)
    pg = [p.to(DEV).requires_grad_(True) for p in packed]
    tg = theta.to(DEV).requires_grad_(True)
    j, kl, z = ops.poe_reparam_kl(tg, pg, [e.to(DEV) for e in eps], with_prior, kl_mask)
    check(j, jr, 1e-5, "joint")
    check(kl, klr, 2e-5, "kl")
    if n_z:
        check(torch.stack(z), zr, 1e-5, "z")
    if kl_mask or n_z:
        tot.backward()
        totg = (kl * gkl.to(DEV)).sum() + ((torch.stack(z) * gz.to(DEV)).sum() if n_z else 0.0)
        totg.backward()
        for e in range(E):
            check(pg[e].grad, pr[e].grad, 5e-5, f"dpacked[{e}]")
        if kl_mask:
            check(tg.grad, tr.grad, 5e-5, "dtheta")


@pytest.mark.parametrize("n_z,B,D", [(2, 128, 32), (3, 7, 9)])
def test_poe_draws_its_own_noise(ops, n_z, B, D):
    """rng=state: the fusion kernel generates the reparameterisation noise itself -- bit-identical z, kl and gradients
    to drawing ops.randn((n_z, B, D), state) first, and the generator advances by exactly one draw."""
    g = torch.Generator().manual_seed(B + D)
    packed = [torch.cat([torch.randn(B, D, generator=g), F.softmax(torch.randn(B, D, generator=g), -1) + 1e-6], -1).to(DEV)
              for _ in range(2)]
    theta = (torch.randn(1, D, generator=g) * 0.3).to(DEV)
    gz = torch.randn(n_z, B, D, generator=g).to(DEV)
    mask = 0b111
    res = []
    for fused in (False, True):
        state = torch.tensor([1234567, 5, 0], dtype=torch.int32, device=DEV)
        pg = [p.clone().requires_grad_(True) for p in packed]
        tg = theta.clone().requires_grad_(True)
        if fused:
            j, kl, z = ops.poe_reparam_kl(tg, pg, n_z, True, mask, rng=state)
        else:
            eps = list(ops.randn((n_z, B, D), state).unbind(0))
            j, kl, z = ops.poe_reparam_kl(tg, pg, eps, True, mask)
        (kl.sum() + (torch.stack(z) * gz).sum()).backward()
        assert state.tolist() == [1234567, 6, 0]
        res.append((kl, torch.stack(z), pg[0].grad, pg[1].grad, tg.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_bce_and_ce(ops):
    g = torch.Generator().manual_seed(3)
    B, Fd = 9, 3 * 64 * 64
    xh = torch.sigmoid(torch.randn(B, Fd, generator=g) * 4).clamp(1e-6, 1 - 1e-6)
    t = torch.rand(B, Fd, generator=g)
    gr = torch.randn(B, generator=g)
    xr = xh.double().requires_grad_(True)
    rr = F.binary_cross_entropy(xr, t.double(), reduction="none").sum(-1)
    rr.backward(gr.double())
    xg = xh.to(DEV).requires_grad_(True)
    r = ops.bce_rowsum(xg, t.to(DEV))
    r.backward(gr.to(DEV))
    check(r, rr, 1e-5, "bce rowsum")
    check(xg.grad, xr.grad, 1e-5, "bce dx")
    e = ops.bce_elem(xh.to(DEV), t.to(DEV))
    check(e, F.binary_cross_entropy(xh.double(), t.double(), reduction="none"), 1e-5, "bce elem")
    # category_ce over time
    B, T, V = 11, 9, 27
    lg = torch.randn(B, T, V, generator=g) * 2
    tg = F.one_hot(torch.randint(0, V, (B, T), generator=g), V).float()
    tg[:, 6:] = 0
    lg[:, 6:] = 0
    for per_v in (True, False):
        lr_ = lg.double().requires_grad_(True)
        ref = -(tg.double() * F.log_softmax(lr_, dim=1)).sum(1)
        ref = ref if per_v else ref.sum(-1)
        gg = torch.randn(*ref.shape, generator=g)
        ref.backward(gg.double())
        lgg = lg.to(DEV).requires_grad_(True)
        out = ops.ce_over_time(lgg, tg.to(DEV), per_v)
        out.backward(gg.to(DEV))
        check(out, ref, 1e-5, f"ce per_v={per_v}")
        check(lgg.grad, lr_.grad, 2e-5, f"ce dlogits per_v={per_v}")


@pytest.mark.parametrize("L,N,d,HN", [(32, 128, 54, 64), (6, 5, 54, 16)])
def test_txt_layer_with_pooled_heads(ops, L, N, d, HN):
    """last encoder layer with the time pooling AND the packed posterior heads in its launch vs layer -> mean -> linear"""
    from multimodal_vae_comparison_amd.models import encoders
    torch.manual_seed(L + N + HN)
    layer = encoders.HipTransformerEncoderLayer(d, 2, 128).to(DEV)
    g = torch.Generator().manual_seed(N)
    x = torch.randn(L, N, d, generator=g).to(DEV)
    lens = torch.randint(1, L + 1, (N,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    hw = (torch.randn(HN, d, generator=g) / math.sqrt(d)).to(DEV)
    hb = (torch.randn(HN, generator=g) * 0.1).to(DEV)
    dh = torch.randn(N, HN, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        for p in layer.parameters():
            p.grad = None
        gw, gb = torch.zeros_like(hw), torch.zeros_like(hb)
        xg = x.clone().requires_grad_(True)
        if fused:
            out = layer(xg, mask, None, time_mean=True, heads=(hw, hb, gw, gb))
        else:
            out = ops.linear(layer(xg, mask, None, time_mean=True), hw, hb, 0, gw, gb)
        out.backward(dh)
        res.append((out.detach(), xg.grad, gw, gb, layer.linear1.weight.grad.clone()))
    for name, a, b in zip(("heads", "dx", "dW_heads", "db_heads", "dW_l1"), *res):
        check(a, b, 2e-5, name)


@pytest.mark.parametrize("B,Fd", [(9, 3 * 64 * 64), (5, 37)])
def test_seeded_losses_write_their_own_backward(ops, B, Fd):
    """ops.ConstSeed: with a known constant upstream gradient the loss kernel writes the logit gradient in the same
    pass (bce on clamp(sigmoid) outputs, category_ce over time) and backward launches nothing -- same numbers as the
    two-kernel path; any other upstream gradient still takes the ordinary backward kernel."""
    g = torch.Generator().manual_seed(B + Fd)
    val = 1.7 / B
    seed = torch.full((B,), val, device=DEV)
    other = torch.randn(B, generator=g).to(DEV)
    y = torch.sigmoid(torch.randn(B, Fd, generator=g) * 6).clamp(1e-6, 1 - 1e-6).to(DEV)
    t = torch.rand(B, Fd, generator=g).to(DEV)
    T, V = 7, 27
    lg = (torch.randn(B, T, V, generator=g) * 2).to(DEV)
    tg = F.one_hot(torch.randint(0, V, (B, T), generator=g), V).float().to(DEV)
    for name, fn, x in (("bce", lambda a: ops.bce_sigmoid_rowsum(a, t), y), ("ce", lambda a: ops.ce_over_time(a, tg, False), lg)):
        a0 = x.clone().requires_grad_(True)
        r0 = fn(a0)
        r0.backward(seed)
        for up in (seed, other):
            a1 = x.clone().requires_grad_(True)
            with ops.ConstSeed(seed, val):
                r1 = fn(a1)
            assert r1.grad_fn.seeded is not None, name
            r1.backward(up)
            assert torch.equal(r1, r0), name
            if up is seed:
                check(a1.grad, a0.grad, 1e-6, f"{name} seeded grad")
            else:
                a2 = x.clone().requires_grad_(True)
                fn(a2).backward(up)
                assert torch.equal(a1.grad, a2.grad), name


@pytest.mark.parametrize("transposed,B,Cin,Cout,Hin,S,P,act,ep", [
    (False, 5, 32, 64, 16, 2, 1, 2, 0), (False, 3, 64, 64, 8, 2, 1, 2, 0), (False, 4, 64, 128, 4, 2, 0, 2, 0),
    (False, 2, 3, 32, 32, 2, 1, 0, 0), (True, 4, 128, 64, 1, 1, 0, 2, 0), (True, 3, 64, 64, 4, 2, 1, 2, 0),
    (True, 3, 64, 32, 8, 2, 1, 2, 0), (True, 2, 32, 3, 16, 2, 1, 2, 7), (True, 2, 16, 5, 6, 2, 1, 1, 7),
    # the 64-channel MFMA instantiations at sizes with several workgroups / ragged last tiles, every plan
    (False, 130, 32, 64, 16, 2, 1, 2, 0), (False, 70, 64, 64, 8, 2, 1, 2, 0), (False, 600, 64, 64, 8, 2, 1, 1, 0),
    (False, 3, 32, 64, 32, 2, 1, 0, 0), (False, 6, 64, 32, 16, 2, 1, 2, 0), (False, 40, 64, 128, 4, 2, 0, 2, 0),
    (True, 260, 64, 64, 4, 2, 1, 2, 0), (True, 65, 64, 32, 8, 2, 1, 2, 0), (True, 900, 64, 32, 8, 2, 1, 2, 0),
    (True, 5, 32, 64, 8, 2, 1, 0, 0), (True, 3, 64, 32, 16, 2, 1, 1, 0), (True, 33, 128, 64, 1, 1, 0, 2, 0),
    (True, 1100, 128, 64, 1, 1, 0, 2, 0),
    # split-bf16 scatter body on the 8 x 8 grid (conv_scatter_b16.inc): 64 reduced channels, odd image counts (the last
    # workgroup holds one image), two output-channel groups, a SiLU' epilogue on the data gradient
    (False, 515, 32, 64, 16, 2, 1, 1, 0), (True, 513, 64, 64, 8, 2, 1, 2, 0)])
def test_conv_generic(ops, transposed, B, Cin, Cout, Hin, S, P, act, ep):
    """ops.conv2d / ops.convT2d (generic kernels, or the MFMA kernels when the shape is theirs) vs torch fp64: the
    SVHN tower layers (channels 64 / 128, k4 s2 p0, k4 s1 p0) and the plain-sigmoid epilogue"""
    g = torch.Generator().manual_seed(Cin * Cout + Hin)
    x = torch.randn(B, Cin, Hin, Hin, generator=g)
    w = torch.randn(*((Cin, Cout) if transposed else (Cout, Cin)), 4, 4, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    f = F.conv_transpose2d if transposed else F.conv2d
    ref = f(_act(xr, act), wr, br, stride=S, padding=P)
    if ep == 7:
        ref = torch.sigmoid(ref)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    if transposed:
        out = ops.convT2d(xg, wg, bg, S, P, act, ep)
    else:
        out = ops.conv2d(xg, wg, bg, S, P, act)
    out.backward(dy.to(DEV))
    check(out, ref, 2e-5, "conv out")
    # the op emits d/d(pre-activation input): compare through act'
    gx = xr.grad
    check(xg.grad, gx, 5e-5, "conv dx")
    check(wg.grad, wr.grad, 5e-5, "conv dw")
    check(bg.grad, br.grad, 5e-5, "conv db")


@pytest.mark.parametrize("rows,trows,C,HW,laplace", [(6, 6, 3, 64, False), (12, 4, 3, 16, True), (10, 5, 1, 49, False)])
def test_lprob_rowsum_nchw_source_with_logit_gradient(ops, rows, trows, C, HW, laplace):
    """the conv decoders' path: loc = sigmoid(logits) stored NCHW, paired with the target as if permuted to NHWC
    (decoders.py:144 + objectives.py:120), K-sample rows against repeated targets, gradient taken w.r.t. the logits"""
    g = torch.Generator().manual_seed(rows + C + HW)
    logits = torch.randn(rows, C, HW, generator=g)
    tgt = torch.rand(trows, C * HW, generator=g)
    gr = torch.randn(rows, generator=g)
    lr_ = logits.clone().requires_grad_(True)
    y = torch.sigmoid(lr_).permute(0, 2, 1).reshape(rows, HW * C)          # what the reference's decoder returns
    t = tgt.repeat(rows // trows, 1)
    dist = torch.distributions.Laplace if laplace else torch.distributions.Normal
    ref = -(dist(y, torch.tensor(0.75)).log_prob(t)).double().sum(-1)
    ref.backward(gr.double())
    raw = torch.sigmoid(logits).to(DEV).requires_grad_(True)               # the layer's epilogue output
    out = ops.lprob_rowsum(raw, tgt.to(DEV), 0.75, laplace, perm_c=C if C > 1 else 0, logit_grad=True)
    out.backward(gr.to(DEV))
    check(out, ref, 1e-5, "rows")
    check(raw.grad, lr_.grad, 1e-4, "d logits")


@pytest.mark.parametrize("transposed,B,Cin,Cout,Hin,S,P", [(False, 9, 32, 64, 16, 2, 1), (False, 9, 64, 64, 8, 2, 1),
                                                          (False, 9, 64, 128, 4, 2, 0), (True, 9, 128, 64, 1, 1, 0),
                                                          (True, 9, 64, 64, 4, 2, 1), (True, 300, 64, 32, 8, 2, 1)])
def test_wide_conv_accumulates_into_preset_gradients(ops, transposed, B, Cin, Cout, Hin, S, P):
    """the SVHN layers with their parameters' preset (flat-buffer) gradient views: split partials go through the
    deferred end-of-backward fold, and the generic (non-MFMA) kernels are never reached"""
    g = torch.Generator().manual_seed(Cin + Cout + Hin)
    x = torch.randn(B, Cin, Hin, Hin, generator=g)
    w = torch.randn(*((Cin, Cout) if transposed else (Cout, Cin)), 4, 4, generator=g) * 0.1
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    f = F.conv_transpose2d if transposed else F.conv2d
    ref = f(torch.relu(xr), wr, br, stride=S, padding=P)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    xg = x.to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV), b.to(DEV)
    gw, gb = torch.ones_like(wd), torch.ones_like(bd)
    generic = ops.ConvGeneric.apply
    ops.ConvGeneric.apply = staticmethod(lambda *a, **k: (_ for _ in ()).throw(AssertionError("generic conv reached")))
    try:
        out = (ops.convT2d(xg, wd, bd, S, P, 2, 0, gw, gb) if transposed else ops.conv2d(xg, wd, bd, S, P, 2, gw, gb))
        out.backward(dy.to(DEV))
    finally:
        ops.ConvGeneric.apply = generic
    torch.cuda.synchronize()
    check(out, ref, 2e-5, "out")
    check(xg.grad, xr.grad, 5e-5, "dx")
    check(gw - 1, wr.grad, 5e-5, "dw (accumulated)")
    check(gb - 1, br.grad, 5e-5, "db (accumulated)")


@pytest.mark.parametrize("B,F_,own,laplace", [(7, 33, False, False), (128, 12288, False, False), (5, 24, True, False),
                                              (6, 40, False, True), (5, 24, True, True)])
def test_lprob_rowsum(ops, B, F_, own, laplace):
    """-log_prob sums vs torch.distributions in fp32 -> double, NaN -> 0 (ReconLoss.lprob); `own`: scale := loc"""
    g = torch.Generator().manual_seed(B + F_)
    loc = torch.randn(B, F_, generator=g)
    if own:
        loc[0, :5] = 0.0                        # exact zeros as in padded steps: 0/0 -> NaN -> dropped
    tgt = torch.randn(B, F_, generator=g)
    tgt[0, :5] = 0.0
    gr = torch.randn(B, generator=g)
    lr_ = loc.clone().requires_grad_(True)
    sc = lr_ if own else torch.tensor(0.75)
    dist = torch.distributions.Laplace if laplace else torch.distributions.Normal
    out = dist(lr_, sc, validate_args=False).log_prob(tgt).double()
    out = torch.where(torch.isnan(out), torch.zeros_like(out), out)
    ref = (-out).sum(1)
    ref.backward(gr.double())
    lg = loc.to(DEV).requires_grad_(True)
    row = ops.lprob_rowsum(lg, tgt.to(DEV), None if own else 0.75, laplace)
    row.backward(gr.to(DEV))
    check(row, ref, 2e-6, "lprob rows")
    rg = torch.where(torch.isnan(lr_.grad), torch.zeros_like(lr_.grad), lr_.grad)
    check(lg.grad, rg, 2e-5, "lprob dloc")


@pytest.mark.parametrize("B,F_", [(5, 24), (128, 128), (64, 12288)])
def test_optimal_sigma_rowsum(ops, B, F_):
    g = torch.Generator().manual_seed(B * 3 + F_)
    loc, tgt, gr = torch.randn(B, F_, generator=g), torch.randn(B, F_, generator=g), torch.randn(B, generator=g)
    lr_ = loc.double().requires_grad_(True)
    ls = -6 + F.softplus(((tgt.double() - lr_) ** 2).mean().sqrt().log() + 6)
    ref = ((((tgt.double() - lr_) / ls.exp()) ** 2).detach() + ls + 0.5 * math.log(2 * math.pi)).sum(1)
    ref.backward(gr.double())
    lg = loc.to(DEV).requires_grad_(True)
    row = ops.optimal_sigma_rowsum(lg, tgt.to(DEV))
    row.backward(gr.to(DEV))
    check(row, ref, 1e-5, "optimal_sigma rows")
    check(lg.grad, lr_.grad, 5e-5, "optimal_sigma dloc")


def test_lincomb_rows(ops):
    g = torch.Generator().manual_seed(4)
    V = torch.randn(5, 130, generator=g)
    W = [[0.5, 1.0, -2.0, 0.25, 3.0], [0.0, 0.0, 1.0, 1.0, 1.0]]
    Vr = V.double().requires_grad_(True)
    ref = torch.tensor(W, dtype=torch.float64) @ Vr.sum(1)
    ref.backward(torch.tensor([1.5, -0.5], dtype=torch.float64))
    Vg = V.to(DEV).requires_grad_(True)
    Va, Vb = Vg[:2], Vg[2:]
    out = ops.lincomb_rows([Va[0], Va[1], Vb], W)          # tuple of scalars, one per row of W
    torch.autograd.backward(list(out), [torch.tensor(1.5, device=DEV), torch.tensor(-0.5, device=DEV)])
    check(torch.stack(out), ref, 1e-5, "lincomb")
    check(Vg.grad, Vr.grad, 1e-6, "lincomb dV")
    # only the first output takes part in backward (the training step: loss.backward(), kld is logged)
    Vg2 = V.to(DEV).requires_grad_(True)
    out2 = ops.lincomb_rows([Vg2], W)
    out2[0].backward(torch.tensor(2.0, device=DEV))
    ref2 = 2.0 * torch.tensor(W[0], dtype=torch.float64)[:, None].expand(5, 130)
    check(Vg2.grad, ref2, 1e-6, "lincomb dV, one output")


@pytest.mark.parametrize("B,T", [(6, 5), (5, 5), (1, 7), (128, 32)])
def test_embed_pe(ops, B, T):
    from multimodal_vae_comparison_amd.models.nn_modules import positional_table
    V = 27
    g = torch.Generator().manual_seed(B + T)
    oh = F.one_hot(torch.randint(0, V, (B, T), generator=g), V).float()
    oh[:, T - 1:] = 0
    emb = torch.randn(V, 2, generator=g)
    pe = positional_table(2, 1000)
    dx = torch.randn(T, B, 2 * V, generator=g)
    er = emb.double().requires_grad_(True)
    x = er[oh.long()]
    p = pe[:B].double()                       # (B,1,2)
    if B == T or B == 1:
        x = x + p
    else:
        x = x.permute(1, 0, 2, 3) + p
    xr = x.contiguous().view(T, B, -1)
    xr.backward(dx.double())
    eg = emb.to(DEV).requires_grad_(True)
    mode = 1 if (B == T or B == 1) else 0
    out = ops.embed_pe(oh.to(DEV), eg, pe.view(-1, 2).to(DEV), mode)
    out.backward(dx.to(DEV))
    check(out, xr, 1e-6, "embed")
    check(eg.grad, er.grad, 2e-5, "demb")
    if mode == 0:      # R passes over the same batch as one call (round 5): row k * B + b == pass k's row b, gradients add up
        R = 3
        eg2 = emb.to(DEV).requires_grad_(True)
        out2 = ops.embed_pe(oh.to(DEV), eg2, pe.view(-1, 2).to(DEV), mode, None, None, R)
        assert tuple(out2.shape) == (T, R * B, 2 * V)
        for k in range(R):
            assert torch.equal(out2[:, k * B:(k + 1) * B], out.detach())
        out2.backward(torch.cat([dx, 2 * dx, -0.5 * dx], 1).to(DEV))
        check(eg2.grad, 2.5 * er.grad, 2e-5, "demb (3 passes)")


@pytest.mark.parametrize("L,N,E,H_", [(5, 6, 54, 2), (32, 128, 54, 2), (9, 7, 32, 2), (64, 3, 16, 2),
                                      (100, 5, 32, 2), (128, 3, 64, 2), (65, 130, 32, 2), (45, 32, 16, 2), (33, 4, 24, 2)])
def test_attention(ops, L, N, E, H_):
    g = torch.Generator().manual_seed(L * N)
    qkv = torch.randn(L, N, 3 * E, generator=g)
    lens = torch.randint(1, L + 1, (N,), generator=g)
    kpm = torch.arange(L)[None, :] >= lens[:, None]          # True = ignore
    do = torch.randn(L, N, E, generator=g)
    qr = qkv.double().requires_grad_(True)
    hd = E // H_
    q, k, v = qr[..., :E], qr[..., E:2 * E], qr[..., 2 * E:]
    q = q.reshape(L, N * H_, hd).transpose(0, 1) / math.sqrt(hd)
    k = k.reshape(L, N * H_, hd).transpose(0, 1)
    v = v.reshape(L, N * H_, hd).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2)).reshape(N, H_, L, L).masked_fill(kpm[:, None, None, :], float("-inf"))
    att = F.softmax(att, -1).reshape(N * H_, L, L)
    ref = torch.bmm(att, v).transpose(0, 1).reshape(L, N, E)
    ref.backward(do.double())
    qg = qkv.to(DEV).requires_grad_(True)
    out = ops.attention(qg, kpm.to(torch.uint8).to(DEV), H_)
    out2 = ops.attention(qg.detach(), (~kpm).to(DEV).view(torch.uint8), H_, mask_is_valid=True)
    check(out2, ref, 2e-5, "attn out (validity-mask form)")
    out.backward(do.to(DEV))
    check(out, ref, 2e-5, "attn out")
    check(qg.grad, qr.grad, 5e-5, "attn dqkv")
    from multimodal_vae_comparison_amd import hipops as H
    was = H.lib().mmvae_attn_t_bwd_set(1)      # ... and the register-form backward
    try:
        qg2 = qkv.to(DEV).requires_grad_(True)
        ops.attention(qg2, kpm.to(torch.uint8).to(DEV), H_).backward(do.to(DEV))
        check(qg2.grad, qr.grad, 5e-5, "attn dqkv (register-form backward)")
    finally:
        H.lib().mmvae_attn_t_bwd_set(was)


@pytest.mark.parametrize("L,N,E,H_,p", [(100, 9, 32, 2, 0.1), (128, 3, 32, 2, 0.3), (65, 4, 24, 2, 0.1), (70, 2, 16, 4, 0.2), (99, 3, 32, 2, 0.2), (67, 130, 32, 2, 0.1),
                                        (40, 5, 32, 2, 0.1), (100, 3, 64, 2, 0.1), (45, 32, 16, 2, 0.1)])
def test_attention_weight_dropout(ops, L, N, E, H_, p):
    """attention with dropout on the softmax weights (nn.MultiheadAttention(dropout=p), the action towers' train mode)
    against torch in fp64 using the very mask the kernel used (mmvae_dropout_mask, element ((n H + h) L + l) S + s):
    the MFMA kernels (64 < L <= 128, head_dim <= 16, with head_dim 12 and 4 too) and the thread-per-row ones"""
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    g = torch.Generator().manual_seed(L * N + E)
    qkv = torch.randn(L, N, 3 * E, generator=g)
    lens = torch.randint(1, L + 1, (N,), generator=g)
    lens[0] = L
    kpm = torch.arange(L)[None, :] >= lens[:, None]
    do = torch.randn(L, N, E, generator=g)
    st = DropoutState().to(DEV)
    slot, call = st.begin()
    drop = st.spec(slot, call, 1, p, "attn")
    mask = ops.dropout_mask(drop, N * H_ * L * L).double().view(N * H_, L, L)
    qr = qkv.double().to(DEV).requires_grad_(True)
    hd = E // H_
    q, k, v = qr[..., :E], qr[..., E:2 * E], qr[..., 2 * E:]
    q = q.reshape(L, N * H_, hd).transpose(0, 1) / math.sqrt(hd)
    k = k.reshape(L, N * H_, hd).transpose(0, 1)
    v = v.reshape(L, N * H_, hd).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2)).reshape(N, H_, L, L).masked_fill(kpm.to(DEV)[:, None, None, :], float("-inf"))
    att = F.softmax(att, -1).reshape(N * H_, L, L) * mask
    ref = torch.bmm(att, v).transpose(0, 1).reshape(L, N, E)
    ref.backward(do.double().to(DEV))
    from multimodal_vae_comparison_amd import hipops as H
    was = H.lib().mmvae_attn_t_bwd_set(0)
    try:
        for form in (0, 1):      # the LDS-tile backward (default) and the register form (csrc/text.hip: attn_t_bwd_kernel)
            H.lib().mmvae_attn_t_bwd_set(form)
            qg = qkv.to(DEV).requires_grad_(True)
            out = ops.attention(qg, kpm.to(torch.uint8).to(DEV), H_, drop=drop)
            out.backward(do.to(DEV))
            check(out, ref, 2e-5, "attn out")
            check(qg.grad, qr.grad, 5e-5, f"attn dqkv (backward form {form})")
    finally:
        H.lib().mmvae_attn_t_bwd_set(was)


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("L,N,d,bcast", [(5, 6, 54, False), (32, 128, 32, True), (1, 300, 8, False), (9, 7, 70, True),
                                         (7, 5, 32, False), (100, 128, 32, False), (33, 1100, 32, True)])
def test_layernorm_residual(ops, L, N, d, bcast, p):
    """LayerNorm(dropout(x) + r) forward / backward (generic kernels and the d = 32 ones: eight threads per row; partial
    row blocks, more rows than one pass of 1024 workgroups), dropout mask extracted from the kernels' generator"""
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    g = torch.Generator().manual_seed(L + N + d)
    x = torch.randn(L, N, d, generator=g)
    r = torch.randn(N, d, generator=g) if bcast else torch.randn(L, N, d, generator=g)
    ga = 1 + 0.1 * torch.randn(d, generator=g)
    be = 0.1 * torch.randn(d, generator=g)
    dy = torch.randn(L, N, d, generator=g)
    drop, mask = None, 1.0
    if p > 0:
        st = DropoutState().to(DEV)
        slot, call = st.begin()
        drop = st.spec(slot, call, 2, p, "drop1")
        mask = ops.dropout_mask(drop, L * N * d).double().view(L, N, d).cpu()
    xr, rr, gr, br = (t.double().requires_grad_(True) for t in (x, r, ga, be))
    ref = F.layer_norm(xr * mask + rr, (d,), gr, br, 1e-5)
    ref.backward(dy.double())
    xg, rg, gg, bg = (t.to(DEV).requires_grad_(True) for t in (x, r, ga, be))
    out = ops.layernorm_residual(xg, rg, gg, bg, drop=drop)
    out.backward(dy.to(DEV))
    check(out, ref, 1e-5, "ln out")
    check(xg.grad, xr.grad, 2e-5, "ln dx")
    check(rg.grad, rr.grad, 2e-5, "ln dr")
    check(gg.grad, gr.grad, 2e-5, "ln dgamma")
    check(bg.grad, br.grad, 2e-5, "ln dbeta")


def test_time_reduce_and_permute_mask(ops):
    g = torch.Generator().manual_seed(9)
    # (7 steps: the serial sum; 100 / 33 / 37 steps: four row groups per workgroup, ragged column counts)
    for L, N, d in ((7, 5, 54), (100, 128, 32), (33, 3, 27), (37, 1, 70)):
        x = torch.randn(L, N, d, generator=g)
        dy = torch.randn(N, d, generator=g)
        xr = x.double().requires_grad_(True)
        xr.mean(0).backward(dy.double())
        xg = x.to(DEV).requires_grad_(True)
        out = ops.mean_over_time(xg)
        out.backward(dy.to(DEV))
        check(out, x.double().mean(0), 1e-6, f"mean {L, N, d}")
        check(xg.grad, xr.grad, 1e-6, f"mean dx {L, N, d}")
    T, B, V = 6, 4, 27
    y = torch.randn(T, B, V, generator=g)
    m = torch.rand(B, T, generator=g) > 0.3
    dz = torch.randn(B, T, V, generator=g)
    yr = y.double().requires_grad_(True)
    ref = yr.permute(1, 0, 2) * m.unsqueeze(-1).double()
    ref.backward(dz.double())
    yg = y.to(DEV).requires_grad_(True)
    o = ops.permute_mask(yg, m.to(torch.uint8).to(DEV))
    o.backward(dz.to(DEV))
    check(o, ref, 1e-6, "permute_mask")
    check(yg.grad, yr.grad, 1e-6, "permute_mask dx")


def test_reduce_segments(ops, hip_lib):
    """dst_s += sum_r src_s[r]: aligned / unaligned lengths and bases, chained segments sharing a destination"""
    import ctypes
    from multimodal_vae_comparison_amd import hipops as H
    g = torch.Generator().manual_seed(3)
    arena = torch.randn(1 << 22, generator=g).to(DEV)
    flat = torch.randn(40000, generator=g).to(DEV)
    ref = flat.clone()
    orig = flat.clone()
    specs = [  # (src offset, dst offset, rows, len, stride)
        (0, 0, 128, 16384, 16416), (16384, 16384, 128, 32, 16416),          # conv partial: weights then bias columns
        (128 * 16416, 16416, 33, 8748, 8748 + 162), (128 * 16416 + 8748, 25164, 33, 162, 8748 + 162),
        (3000001, 30001, 7, 1001, 1003),                                       # nothing aligned
        (3100000, 0, 5, 16384, 16384),                                         # chained onto segment 0 (same dst, len)
        (3300000, 33000, 1, 64, 64),
        # >= 128 rows: folded by the narrow blocks (64 columns x 16 row groups) -- a 3-channel conv layer's 512 partial
        # rows, and an unaligned one
        (2400100, 34000, 512, 1568, 1600), (3500001, 36001, 130, 77, 79)]
    t = H.ReduceSegments()
    for j, (so, do, r, ln, sd) in enumerate(specs):
        t.src[j], t.dst[j] = arena.data_ptr() + 4 * so, flat.data_ptr() + 4 * do
        t.rows[j], t.len[j], t.stride[j] = r, ln, sd
        rows = torch.stack([arena[so + k * sd: so + k * sd + ln] for k in range(r)])
        ref[do:do + ln] += rows.double().sum(0).float()
    t.n = len(specs)
    H.check(hip_lib.mmvae_reduce_segments(ctypes.byref(t), H.stream()), "mmvae_reduce_segments")
    check(flat, ref, 2e-6, "reduce_segments")
    # the same fold once more with the ELBO assembly riding along as one extra workgroup
    before = flat.clone()
    blocks = [torch.randn(130, generator=g).to(DEV), torch.randn(3, 130, generator=g).to(DEV)]
    W = [[0.5, 0.1, 0.2, 0.3], [0.0, 1.0, 1.0, 1.0]]
    tail = ops.lincomb_rows_args(blocks, W)
    rp, wf, out, n, B, k = tail["args"]
    H.check(hip_lib.mmvae_reduce_segments_lincomb(ctypes.byref(t), ctypes.byref(rp), wf, H.ptr(out), n, B, k, H.stream()),
            "mmvae_reduce_segments_lincomb")
    rows = torch.cat([blocks[0][None], blocks[1]]).double().sum(1)
    check(out, torch.tensor(W, dtype=torch.float64, device=DEV) @ rows, 1e-5, "lincomb rider")
    check(flat, before.double() + (ref.double() - orig.double()), 4e-6, "reduce_segments with rider")


@pytest.mark.parametrize("rider", [False, True])
def test_adam_fold_flat_is_fold_then_adam(ops, hip_lib, rider):
    """mmvae_adam_fold_flat == mmvae_reduce_segments[_lincomb] followed by mmvae_adam_amsgrad_flat(step=-1), BIT for bit
    (parameters, both moments, the running maximum, the cleared gradient buffer, the device step block), over three
    steps; aligned / unaligned / chained segments, gaps of every alignment between them, a segment at the very end"""
    import ctypes
    from multimodal_vae_comparison_amd import hipops as H
    g = torch.Generator().manual_seed(5)
    n = 40000
    arena = torch.randn(1 << 22, generator=g).to(DEV)
    specs = [  # (src offset, dst offset, rows, len, stride), unsorted on purpose
        (128 * 16416, 16416, 33, 8748, 8748 + 162), (0, 0, 128, 16384, 16416), (16384, 16384, 128, 32, 16416),
        (128 * 16416 + 8748, 25164, 33, 162, 8748 + 162), (3000001, 30001, 7, 1001, 1003),
        (3100000, 0, 5, 16384, 16384), (3300000, 33000, 1, 64, 64), (3400000, n - 13, 9, 13, 16),
        (2400100, 34000, 512, 1568, 1600), (3500001, 36001, 130, 77, 79)]      # narrow-block segments (>= 128 rows)
    state = {}
    split = 30000      # "c": the step in two launches (mmvae_adam_fold_range): [split, n) first, then [0, split) closes it
    for name in ("a", "b", "c"):
        gg = torch.Generator().manual_seed(6)
        state[name] = {"p": torch.randn(n, generator=gg).to(DEV), "m": torch.zeros(n, device=DEV),
                       "v": torch.zeros(n, device=DEV), "x": torch.zeros(n, device=DEV), "g": torch.zeros(n, device=DEV),
                       "step": torch.zeros(6, dtype=torch.int32, device=DEV)}
    blocks = [torch.randn(130, generator=g).to(DEV), torch.randn(3, 130, generator=g).to(DEV)]
    W = [[0.5, 0.1, 0.2, 0.3], [0.0, 1.0, 1.0, 1.0]]
    for it in range(3):
        direct = (torch.randn(n, generator=g) * (torch.rand(n, generator=g) < 0.5)).to(DEV)   # what kernels wrote straight to g
        arena.mul_(0.9).add_(0.01 * it)
        outs = {}
        for name, st in state.items():
            st["g"].copy_(direct)
            t = H.ReduceSegments()
            for j, (so, do, r, ln, sd) in enumerate(specs):
                t.src[j], t.dst[j] = arena.data_ptr() + 4 * so, st["g"].data_ptr() + 4 * do
                t.rows[j], t.len[j], t.stride[j] = r, ln, sd
            t.n = len(specs)
            tail = ops.lincomb_rows_args(blocks, W) if rider else None
            if name == "a":
                if rider:
                    rp, wf, out, nr, B, k = tail["args"]
                    H.check(hip_lib.mmvae_reduce_segments_lincomb(ctypes.byref(t), ctypes.byref(rp), wf, H.ptr(out), nr, B,
                                                                  k, H.stream()), "fold")
                else:
                    H.check(hip_lib.mmvae_reduce_segments(ctypes.byref(t), H.stream()), "fold")
                ops.adam_amsgrad_flat(st["p"], st["g"], st["m"], st["v"], st["x"], 1e-3, 0.9, 0.999, 1e-8, -1, st["step"],
                                      0.5, True)
            elif name == "b":
                ops.adam_fold_flat(st["p"], st["g"], st["m"], st["v"], st["x"], 1e-3, 0.9, 0.999, 1e-8, st["step"], 0.5,
                                   True, {"table": t, "tail": tail})
            else:
                step_before = int(st["step"][0])
                ops.adam_fold_range(st["p"], st["g"], st["m"], st["v"], st["x"], split, n, False, 1e-3, 0.9, 0.999, 1e-8,
                                    st["step"], 0.5, True, t)
                torch.cuda.synchronize()
                assert int(st["step"][0]) == step_before and float(st["g"][split:].abs().max()) == 0.0
                # (the first range once with a table that only holds ITS segments, once with the full table)
                if it == 1:
                    t1 = H.ReduceSegments()
                    keep = [sp for sp in specs if sp[1] < split]
                    for j, (so, do, r, ln, sd) in enumerate(keep):
                        t1.src[j], t1.dst[j] = arena.data_ptr() + 4 * so, st["g"].data_ptr() + 4 * do
                        t1.rows[j], t1.len[j], t1.stride[j] = r, ln, sd
                    t1.n = len(keep)
                    t = t1
                ops.adam_fold_range(st["p"], st["g"], st["m"], st["v"], st["x"], 0, split, True, 1e-3, 0.9, 0.999, 1e-8,
                                    st["step"], 0.5, True, t, tail=tail)
            outs[name] = tail["args"][2].clone() if rider else None
        torch.cuda.synchronize()
        for other in ("b", "c"):
            for k in ("p", "m", "v", "x", "g", "step"):
                da, db = state["a"][k], state[other][k]
                if not torch.equal(da, db):
                    bad = (da != db).nonzero().flatten()
                    raise AssertionError(f"step {it} ({other}): {k} differs at {bad.numel()} elements, first {bad[:8].tolist()}, "
                                         f"last {bad[-4:].tolist()}, max |d| {float((da.double() - db.double()).abs().max()):.3e}")
            if rider:
                assert torch.equal(outs["a"], outs[other])
    assert int(state["b"]["step"][0]) == 3 and float(state["b"]["g"].abs().max()) == 0.0
    # a segment that straddles the range boundary is refused; a range without segments is plain Adam over it
    st = state["c"]
    tb = H.ReduceSegments()
    tb.src[0], tb.dst[0], tb.rows[0], tb.len[0], tb.stride[0] = arena.data_ptr(), st["g"].data_ptr() + 4 * (split - 8), 2, 16, 16
    tb.n = 1
    with pytest.raises(RuntimeError):
        ops.adam_fold_range(st["p"], st["g"], st["m"], st["v"], st["x"], split, n, False, 1e-3, 0.9, 0.999, 1e-8,
                            st["step"], 0.5, True, tb)
    p_before = st["p"].clone()
    st["g"][split:split + 100] = 1.0
    ops.adam_fold_range(st["p"], st["g"], st["m"], st["v"], st["x"], split, n, False, 1e-3, 0.9, 0.999, 1e-8, st["step"],
                        0.5, True, None)
    torch.cuda.synchronize()
    assert torch.equal(st["p"][:split], p_before[:split]) and not torch.equal(st["p"][split:split + 100], p_before[split:split + 100])
    # destinations outside the flat buffer / overlapping ranges are refused
    t = H.ReduceSegments()
    t.src[0], t.dst[0], t.rows[0], t.len[0], t.stride[0] = arena.data_ptr(), state["b"]["g"].data_ptr() + 4 * (n - 8), 2, 16, 16
    t.n = 1
    st = state["b"]
    rc = hip_lib.mmvae_adam_fold_flat(H.ptr(st["p"]), H.ptr(st["g"]), H.ptr(st["m"]), H.ptr(st["v"]), H.ptr(st["x"]), n,
                                      1e-3, 0.9, 0.999, 1e-8, H.ptr(st["step"]), 1.0, 1, ctypes.byref(t), None, None, None,
                                      0, 0, 0, H.stream())
    assert rc == 1      # MMVAE_ERR_ARG


@pytest.mark.parametrize("core", ["split_bf16", "fp32"])
@pytest.mark.parametrize("M,FF,p", [(600, 1024, 0.1), (70, 128, 0.0), (12800, 1024, 0.1), (33, 96, 0.3), (1, 32, 0.5)])
def test_ffn32_fused_matches_torch_and_op_by_op(ops, monkeypatch, M, FF, p, core):
    """csrc/ffn.hip / ffn_b16.inc: linear2(dropout(gelu(linear1(x)))) with d_model 32 in three fused launches (forward,
    data gradient, weight gradients; on split-bf16 MFMA -- what ships -- and on fp32 MFMA) against torch in fp64 with the
    very dropout mask the kernel used (extracted with mmvae_dropout_mask: same element index row * FF + col as the
    op-by-op activation-dropout pass), ragged row counts (partial row blocks, partial row slices), FF not a multiple of the
    4-wave stride.  Same bars for both cores: 2e-6 / 5e-6 of the tensor maximum (a two-term split would sit at ~1e-5)."""
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    monkeypatch.setattr(ops, "FFN32_SPLIT_BF16", core == "split_bf16")
    g = torch.Generator().manual_seed(M + FF)
    x = torch.randn(M, 32, generator=g).to(DEV).requires_grad_(True)
    w1 = (torch.randn(FF, 32, generator=g) * 0.3).to(DEV).requires_grad_(True)
    b1 = (torch.randn(FF, generator=g) * 0.3).to(DEV).requires_grad_(True)
    w2 = (torch.randn(32, FF, generator=g) * 0.1).to(DEV).requires_grad_(True)
    b2 = torch.randn(32, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(M, 32, generator=g).to(DEV)
    drop = None
    mask = torch.ones(M, FF, dtype=torch.float64, device=DEV)
    if p > 0:
        st = DropoutState().to(DEV)
        slot, call = st.begin()
        drop = st.spec(slot, call, 3, p, "ffn")
        mask = ops.dropout_mask(drop, M * FF).double().view(M, FF)
        assert 0.5 * (1 - p) < float((mask > 0).double().mean()) <= 1.0
    y = ops.ffn32(x, w1, b1, w2, b2, drop)
    y.backward(dy)
    xr, w1r, b1r, w2r, b2r = (t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    ref = (F.gelu(xr @ w1r.t() + b1r) * mask) @ w2r.t() + b2r
    ref.backward(dy.double())
    check(y, ref, 2e-6, "y")
    for a, r, name in ((x, xr, "dx"), (w1, w1r, "dW1"), (b1, b1r, "db1"), (w2, w2r, "dW2"), (b2, b2r, "db2")):
        check(a.grad, r.grad, 5e-6, name)
    # ... and the op-by-op kernels on the same inputs and mask
    x2, w12, b12, w22, b22 = (t.detach().clone().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    h = ops.linear(x2, w12, b12, 0)
    h = ops.dropout_act(h, 3, drop) if drop is not None else ops.dropout_act(h, 3, None)
    y2 = ops.linear(h, w22, b22, 0)
    y2.backward(dy)
    check(y, y2, 2e-6, "y vs op-by-op")
    check(x.grad, x2.grad, 5e-6, "dx vs op-by-op")
    check(w1.grad, w12.grad, 5e-6, "dW1 vs op-by-op")


@pytest.mark.parametrize("L,N,H,hd,p", [(100, 128, 2, 16, 0.1), (7, 5, 2, 16, 0.3), (33, 3, 4, 12, 0.0)])
def test_head_bcast_dropout_matches_torch(ops, L, N, H, hd, p):
    """cross-attention over a length-1 memory (models/decoders.py: the action decoder's layers): out[l, n, c] = v[n, c] *
    mask[(n H + c / hd) L + l], backward the masked sum over time; mask extracted from the kernels' own generator"""
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    g = torch.Generator().manual_seed(L + N)
    v = torch.randn(N, H * hd, generator=g).to(DEV).requires_grad_(True)
    dout = torch.randn(L, N, H * hd, generator=g).to(DEV)
    drop, mask = None, torch.ones(N, H, L, dtype=torch.float64, device=DEV)
    if p > 0:
        st = DropoutState().to(DEV)
        slot, call = st.begin()
        drop = st.spec(slot, call, 5, p, "attn")
        mask = ops.dropout_mask(drop, N * H * L).double().view(N, H, L)
    out = ops.head_bcast_dropout(v, L, H, drop)
    out.backward(dout)
    m = mask.permute(2, 0, 1).unsqueeze(-1).expand(L, N, H, hd).reshape(L, N, H * hd)
    vr = v.detach().double().requires_grad_(True)
    ref = vr.unsqueeze(0) * m
    ref.backward(dout.double())
    check(out, ref, 1e-6, "out")
    check(v.grad, vr.grad, 2e-6, "dv")


@pytest.mark.parametrize("dec,d,ff,L,N", [(False, 32, 256, 40, 6), (True, 32, 64, 37, 5), (False, 54, 128, 34, 4)])
@pytest.mark.parametrize("train", [True, False])
def test_residual_handoff_equals_autograd_add(ops, monkeypatch, dec, d, ff, L, N, train):
    """ops.ResidualGrad: the LayerNorm's residual gradient added inside the sub-layer's first backward kernel (MMVAE_EP_ADD_AUX
    in the in-projection's data gradient, dx_add in the fused feed-forward block) against autograd's own elementwise add, on
    unfused post-norm layers (d = 32: fused feed-forward; d = 54: plain linears -- only the attention block hands off), in
    train mode with identical dropout masks and in eval mode"""
    from multimodal_vae_comparison_amd.models import decoders, encoders
    torch.manual_seed(d + L)
    layer = (decoders.HipTransformerDecoderLayer if dec else encoders.HipTransformerEncoderLayer)(d, 2, ff).to(DEV)
    monkeypatch.setattr(encoders, "FUSED_TXT_LAYERS", False)
    x0 = torch.randn(L, N, d, device=DEV)
    mem = torch.randn(N, d, device=DEV) if dec else None
    valid = torch.ones(N, L, dtype=torch.uint8, device=DEV)
    valid[0, L - 3:] = 0
    dy = torch.randn(L, N, d, device=DEV)
    res = {}
    st = encoders.DropoutState().to(DEV)
    st0 = st.state.clone()
    for on in (False, True):
        monkeypatch.setattr(ops, "RESIDUAL_HANDOFF", on)
        ds = None
        if train:
            st.state.copy_(st0)                        # the same seed and counters in both passes -> the same masks
            st.reset_calls()
            slot, call = st.begin()
            names = ("attn", "drop1", "xattn", "drop2", "ffn", "drop3") if dec else ("attn", "drop1", "ffn", "drop2")
            ds = {nm: st.spec(slot, call, 1 + i, 0.1, nm) for i, nm in enumerate(names)}
        for p_ in layer.parameters():
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        y = layer(x, mem, valid, ds) if dec else layer(x, valid, ds)
        y.backward(dy)
        res[on] = (y.detach().clone(), x.grad.clone(), {k: p_.grad.clone() for k, p_ in layer.named_parameters()})
    assert torch.equal(res[False][0], res[True][0])
    check(res[True][1], res[False][1], 2e-6, "dx with the hand-off vs autograd's add")
    for k in res[False][2]:
        check(res[True][2][k], res[False][2][k], 2e-6, k)


@pytest.mark.parametrize("self_counting", [False, True])
def test_adam_amsgrad_flat_matches_torch(ops, self_counting):
    """self_counting: step = -1, the kernel bumps the device step counter itself (what FlatAdam uses)"""
    g = torch.Generator().manual_seed(11)
    n = 10007
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, amsgrad=True)
    p = p0.to(DEV)
    # the flat kernel needs 16-byte aligned bases; torch allocations are
    m, v, vm = (torch.zeros(n, device=DEV) for _ in range(3))
    step_dev = torch.zeros(6, dtype=torch.int32, device=DEV)      # {count, ticket, beta1^count, beta2^count}
    for it in range(1, 6):
        gr = torch.randn(n, generator=g) * (0.1 if it != 3 else 10.0)
        pr.grad = gr.clone()
        opt.step()
        gd = gr.to(DEV)
        if self_counting:
            ops.adam_amsgrad_flat(p, gd, m, v, vm, 1e-3, 0.9, 0.999, 1e-8, -1, step_dev, 1.0, True)
            assert step_dev[:2].tolist() == [it, 0]
        else:
            ops.step_inc(step_dev)
            ops.adam_amsgrad_flat(p, gd, m, v, vm, 1e-3, 0.9, 0.999, 1e-8, 0, step_dev, 1.0, True)
        assert float(gd.abs().max()) == 0.0
        check(p, pr.detach(), 2e-6, f"adam step {it}")


@pytest.mark.parametrize("dec,L,N,d,train", [(False, 32, 128, 54, False), (False, 32, 16, 54, True), (False, 5, 6, 54, True),
                                             (True, 32, 128, 32, False), (True, 32, 16, 32, True), (True, 9, 7, 32, True),
                                             (False, 6, 5, 32, True), (True, 6, 5, 54, True),
                                             (False, 32, 128, 54, "mean"), (False, 7, 6, 54, "mean"),
                                             (True, 8, 32, 16, True), (True, 5, 3, 16, False),
                                             (True, 32, 24, 24, True), (True, 7, 5, 24, False)])
@pytest.mark.parametrize("family", ["wave", "wave_fwd", "workgroup"])
def test_txt_layer_fused_matches_op_by_op(ops, hip_lib, family, dec, L, N, d, train):
    """csrc/txtlayer.hip / csrc/txtwave.hip (one launch per layer and direction; family: wave-per-sequence kernels in
    both directions | forward only | the workgroup-per-sequence kernels) against the op-by-op kernels (each checked
    against torch above): outputs, input gradients and every parameter gradient, with identical dropout masks."""
    hip_lib.mmvae_txt_layer_plan(*{"wave": (1, 0, 0), "wave_fwd": (1, 1 << 30, 1 << 30), "workgroup": (0, 1 << 30, 1 << 30)}[family])
    from multimodal_vae_comparison_amd.models import decoders, encoders
    from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
    pooled = train == "mean"      # encoder layer with the time pooling folded in (train mode)
    torch.manual_seed(L * N + d + int(dec))
    layer = (decoders.HipTransformerDecoderLayer if dec else encoders.HipTransformerEncoderLayer)(d, 2, 128).to(DEV)
    for p in layer.parameters():          # non-trivial LayerNorm parameters and biases
        p.data.add_(0.05 * torch.randn_like(p))
    g = torch.Generator().manual_seed(L + N)
    x = torch.randn(L, N, d, generator=g).to(DEV)
    mem = torch.randn(N, d, generator=g).to(DEV)
    lens = torch.randint(1, L + 1, (N,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    dy = (torch.randn(N, d, generator=g) if pooled else torch.randn(L, N, d, generator=g)).to(DEV)
    st = DropoutState().to(DEV)
    sites = ("attn", "drop1", "xattn", "drop2", "ffn", "drop3") if dec else ("attn", "drop1", "ffn", "drop2")
    ds = {k: st.spec(0, 0, i + 1, 0.1, k) for i, k in enumerate(sites)} if train else None

    def run(fused):
        encoders.FUSED_TXT_LAYERS = fused
        for p in layer.parameters():
            p.grad = None
        xg, mg = x.clone().requires_grad_(True), mem.clone().requires_grad_(True)
        out = layer(xg, mg, mask, ds) if dec else layer(xg, mask, ds, time_mean=pooled)
        out.backward(dy)
        return out.detach(), xg.grad, (mg.grad if dec else None), {k: p.grad.clone() for k, p in layer.named_parameters()
                                                                    if p.grad is not None}

    try:
        assert ops.txt_layer_supported(L, d, 128, 2, dec)
        yf, dxf, dmf, gf = run(True)
        yu, dxu, dmu, gu = run(False)
    finally:
        encoders.FUSED_TXT_LAYERS = True
        hip_lib.mmvae_txt_layer_plan(1, 384, 384)
    check(yf, yu, 2e-5, "layer out")
    check(dxf, dxu, 5e-5, "layer dx")
    if dec:
        check(dmf, dmu, 5e-5, "layer dmem")
    assert set(gf) == set(gu), set(gf) ^ set(gu)
    for k in gu:
        check(gf[k], gu[k], 5e-5, f"grad {k}")


# ---------------------------------------------------------------------------------------------
# ResNet-50 tower: global average pooling (csrc/rconv.hip)
# ---------------------------------------------------------------------------------------------
def _nhwc(t):        # (B,C,H,W) -> (B*H*W, C)
    B, C, Hh, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * Hh * W, C).contiguous()


def test_avgpool_matches_torch(ops):
    """(the tower's other pieces -- stem, bottlenecks, BatchNorms, max pooling -- are tests/test_rconv_gpu.py)"""
    g = torch.Generator().manual_seed(3)
    B = 5
    x2 = torch.randn(B, 2048, 2, 2, generator=g)
    x2r = x2.double().requires_grad_(True)
    ref2 = F.adaptive_avg_pool2d(torch.relu(x2r), 1).flatten(1)
    d2 = torch.randn(ref2.shape, generator=g)
    ref2.backward(d2.double())
    x2g = _nhwc(x2).to(DEV).requires_grad_(True)
    out2 = ops.AvgPoolGlobal.apply(x2g, B, 4, 2)
    out2.backward(d2.to(DEV))
    check(out2, ref2, 1e-6, "avgpool")
    check(x2g.grad, _nhwc(x2r.grad), 1e-6, "avgpool dx")


@pytest.mark.parametrize("rows,D,FF,small", [(4096, 32, 128, 128), (4096, 54, 128, 0), (1536, 16, 128, 48), (300, 24, 40, 7)])
def test_linear_bwd_weight_batch_is_the_separate_calls(hip_lib, rows, D, FF, small):
    """mmvae_linear_bwd_weight_batch: the weight gradients a fused text layer leaves behind (4 for the encoder, 6 for
    the decoder, the last over the batch rows only) in one launch -- every partial and every directly written gradient
    bit-identical to the one-call-per-job form, and their sum against float64."""
    import ctypes
    from multimodal_vae_comparison_amd import hipops as H
    lib = H.lib()
    g = torch.Generator().manual_seed(rows + D)
    shapes = [(rows, 3 * D, D), (rows, D, D), (rows, FF, D), (rows, D, FF)]
    if small:
        shapes += [(rows, D, D), (small, D, D)]
    prob = [(torch.randn(M, N, generator=g).to(DEV), torch.randn(M, K, generator=g).to(DEV)) for M, N, K in shapes]
    st = torch.cuda.current_stream().cuda_stream

    def buffers():
        out = []
        for (M, N, K) in shapes:
            nz = lib.mmvae_linear_bwd_weight_splits(M, N, K)
            nws = lib.mmvae_linear_bwd_weight_ws_floats(M, N, K)
            out.append((torch.full((N, K), 0.5, device=DEV), torch.full((N,), 0.25, device=DEV),
                        torch.zeros(max(nws, 1), device=DEV), nz))
        return out

    one, many = buffers(), buffers()
    for (dy, x), (dw, db, ws, nz), (M, N, K) in zip(prob, one, shapes):
        assert lib.mmvae_linear_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N,
                                           K, K, H.ACT_NONE, H.ACC_DEFER, st) == 0
    arr = (H.WgradJob * len(shapes))()
    for j, (dy, x), (dw, db, ws, nz), (M, N, K) in zip(arr, prob, many, shapes):
        j.dy, j.x, j.dw, j.db, j.ws = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr()
        j.M, j.N, j.K, j.ldx, j.x_act, j.accumulate = M, N, K, K, H.ACT_NONE, H.ACC_DEFER
    assert lib.mmvae_linear_bwd_weight_batch(ctypes.cast(arr, ctypes.c_void_p), len(shapes), st) == 0
    torch.cuda.synchronize()
    for (dy, x), (dw1, db1, ws1, nz), (dw2, db2, ws2, _), (M, N, K) in zip(prob, one, many, shapes):
        assert torch.equal(ws1, ws2) and torch.equal(dw1, dw2) and torch.equal(db1, db2), (M, N, K)
        ref_w, ref_b = dy.double().t() @ x.double(), dy.double().sum(0)
        if nz > 1:      # deferred: nz partial slices [dW | db]
            got_w = ws2[:nz * N * K].view(nz, N, K).double().sum(0)
            got_b = ws2[nz * N * K:nz * (N * K + N)].view(nz, N).double().sum(0)
        else:           # one split: accumulated straight into the (preset) gradient
            got_w, got_b = dw2.double() - 0.5, db2.double() - 0.25
        check(got_w, ref_w, 5e-5, f"dw {M, N, K}")
        check(got_b, ref_b, 5e-5, f"db {M, N, K}")


@pytest.mark.parametrize("rows,D,FF,small", [(4096, 32, 128, 128), (4096, 54, 128, 0), (1536, 16, 128, 48), (300, 24, 40, 7),
                                             (32000, 54, 128, 1000), (37, 54, 128, 3), (1, 32, 128, 1)])
def test_txt_wgrad_matches_float64(hip_lib, rows, D, FF, small):
    """mmvae_txt_wgrad (csrc/twgrad.hip): the weight and bias gradients behind one fused text layer in one launch; the
    sum of every job's partial rows against float64 (ragged row counts, widths that are no multiple of 32 or 64,
    the cross-attention value job over the batch rows only), and nothing written outside a job's workspace."""
    import ctypes
    from multimodal_vae_comparison_amd import hipops as H
    lib = H.lib()
    g = torch.Generator().manual_seed(rows + D)
    shapes = [(rows, 3 * D, D), (rows, D, D), (rows, FF, D), (rows, D, FF)]
    if small:
        shapes += [(rows, D, D), (small, D, D)]
    prob = [(torch.randn(M, N, generator=g).to(DEV), torch.randn(M, K, generator=g).to(DEV)) for M, N, K in shapes]
    st = torch.cuda.current_stream().cuda_stream
    arr = (H.TxtWgradJob * len(shapes))()
    wss = []
    for j, (dy, x), (M, N, K) in zip(arr, prob, shapes):
        assert lib.mmvae_txt_wgrad_supported(M, N, K) == 1
        nz = lib.mmvae_txt_wgrad_splits(M, N, K)
        nws = lib.mmvae_txt_wgrad_ws_floats(M, N, K)
        assert nws == nz * ((N * K + N + 1) // 2 * 2)      # (row pitch: N * K + N rounded up to even)
        ws = torch.full((nws + 64,), float("nan"), device=DEV)
        ws[nws:] = 7.0
        wss.append((ws, nz, nws))
        j.dy, j.x, j.ws, j.M, j.N, j.K = dy.data_ptr(), x.data_ptr(), ws.data_ptr(), M, N, K
    assert lib.mmvae_txt_wgrad(ctypes.cast(arr, ctypes.c_void_p), len(shapes), st) == 0
    torch.cuda.synchronize()
    for (dy, x), (ws, nz, nws), (M, N, K) in zip(prob, wss, shapes):
        part = ws[:nws].view(nz, nws // nz)                         # row z: [N * K weight sums | N bias sums | pad]
        assert bool(torch.isfinite(part[:, :N * K + N]).all()), (M, N, K)      # every partial row written
        assert bool((ws[nws:] == 7.0).all()), (M, N, K)
        ref_w, ref_b = dy.double().t() @ x.double(), dy.double().sum(0)
        got_w = part[:, :N * K].reshape(nz, N, K).double().sum(0)
        got_b = part[:, N * K:N * K + N].double().sum(0)
        check(got_w, ref_w, 5e-5, f"dw {M, N, K}")
        check(got_b, ref_b, 5e-5, f"db {M, N, K}")
    assert lib.mmvae_txt_wgrad_supported(64, 35, 32) == 0 and lib.mmvae_txt_wgrad_supported(64, 31, 33) == 0


def test_dropout_advance_many(ops, hip_lib):
    """mmvae_dropout_advance_many == mmvae_dropout_advance(state, 0) on every state; duplicates are refused"""
    import ctypes
    from multimodal_vae_comparison_amd import hipops as H
    a = [torch.tensor([7 + i, 10 * i, 0, 5, 0] + [0] * 13, dtype=torch.int32, device=DEV) for i in range(5)]
    b = [t.clone() for t in a]
    for t in a:
        ops.dropout_advance(t, 0)
    ops.dropout_advance_many(b)
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), i
        assert int(y[1]) == 10 * i + 1 and int(y[2]) == 10 * i + 1 and int(y[3]) == 5 and int(y[0]) == 7 + i
    arr = (ctypes.c_void_p * 2)(b[0].data_ptr(), b[0].data_ptr())
    assert hip_lib.mmvae_dropout_advance_many(ctypes.cast(arr, ctypes.c_void_p), 2, H.stream()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,V,Tk", [(45, 32, 27, 8), (32, 7, 5, 31), (9, 3, 4, 1)])
def test_permute_mask_head_is_permute_mask_then_slice(hip_lib, T, B, V, Tk):
    """mmvae_permute_mask_head_fwd / _bwd (the decoder's permute(1,0,2) * mask, decoders.py:720-722, and the slice to the
    target's mask length, objectives.py:30-52, as one launch): bit-equal to the two steps; the gradient of the steps
    beyond Tk is an exact zero, and a padded step stays a masked WRITE (no NaN leaks through)."""
    from multimodal_vae_comparison_amd import ops
    torch.manual_seed(3)
    x = torch.randn(T, B, V, device=DEV)
    x[0, 0, 0] = float("nan")
    mask = torch.rand(B, T, device=DEV) > 0.3
    mask[0, 0] = False
    mask_u8 = ops.as_u8(mask)
    x1 = x.clone().requires_grad_(True)
    y = ops.permute_mask(x1, mask_u8, Tk)
    ref_in = x.clone().requires_grad_(True)
    ref = torch.where(mask.unsqueeze(-1), ref_in.permute(1, 0, 2), torch.zeros((), device=DEV))[:, :Tk]
    assert y.shape == (B, Tk, V) and torch.equal(y, ref)
    g = torch.randn(B, Tk, V, device=DEV)
    y.backward(g)
    ref.backward(g)
    assert torch.equal(x1.grad, ref_in.grad)
    assert float(x1.grad[Tk:].abs().sum()) == 0.0


@pytest.mark.gpu
def test_rows_fan_is_cat_and_repeat_with_summed_gradients(hip_lib):
    """ops.RowsFan (mmvae_rows_fan_fwd / _bwd): the decoders' input batches from the latent samples in one launch -- PoE's
    row blocks per decoder call and DMVAE's [shared | private] rows -- bit-equal to torch.cat / repeat, and every sample's
    gradient the sum over the blocks that read it (against autograd on the torch ops, 1e-6: the order of the additions may
    differ); an output without a gradient contributes nothing."""
    from multimodal_vae_comparison_amd import ops
    torch.manual_seed(5)
    B, D, P = 37, 20, 10
    z = [torch.randn(B, D, device=DEV) for _ in range(4)]
    zp = [torch.randn(B, P, device=DEV) for _ in range(2)]
    srcs = z + zp
    plan = [(3, D, [(0, 0, 0), (1, 1, 0), (2, 2, 0)]),                                        # PoE: three subsets' samples
            (1, D, [(0, 0, 0)]),
            (3, D + P, [(1, 0, 0), (3, 1, 0), (2, 2, 0), (4, 0, D), (4, 1, D), (4, 2, D)]),    # DMVAE: [shared | private]
            (2, D + P, [(3, 0, 0), (0, 1, 0), (5, 0, D), (5, 1, D)])]
    assert ops.RowsFan.supported(plan, srcs)
    a = [t.clone().requires_grad_(True) for t in srcs]
    b = [t.clone().requires_grad_(True) for t in srcs]
    outs = ops.rows_fan(plan, a)
    ref = [torch.cat([b[0], b[1], b[2]], 0), b[0] * 1.0,
           torch.cat([torch.cat([b[1], b[3], b[2]], 0), b[4].repeat(3, 1)], -1),
           torch.cat([torch.cat([b[3], b[0]], 0), b[5].repeat(2, 1)], -1)]
    for o, r in zip(outs, ref):
        assert o.shape == r.shape and torch.equal(o, r)
    gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward([outs[0], outs[2], outs[3]], [gs[0], gs[2], gs[3]])      # (output 1 gets no gradient)
    torch.autograd.backward([ref[0], ref[2], ref[3]], [gs[0], gs[2], gs[3]])
    for i, (x, y) in enumerate(zip(a, b)):
        check(x.grad, y.grad.double(), 1e-6, f"d source {i}")
    too_many = [(1, D, [(0, 0, 0)])] * 17
    assert not ops.RowsFan.supported(too_many, srcs[:1])


@pytest.mark.gpu
def test_poe_column_ranges_share_one_gradient_tensor(hip_lib, monkeypatch):
    """Several fusion calls over column ranges of ONE head output (DMVAE: joint over the shared columns, each modality's
    shared and private posteriors): with ops.GradReducer.share_packed_grads the calls' backward passes accumulate into one
    gradient tensor (mmvae_poe_reparam_kl_bwd_acc) instead of a zero-filled tensor per call + autograd's additions -- the head
    output's gradient equal to the per-call form (2e-7: the order of two additions), twice in a row (the registry of
    shared tensors lives for one backward pass only)."""
    from multimodal_vae_comparison_amd import ops
    torch.manual_seed(11)
    B, D, P = 33, 20, 10
    theta = torch.randn(1, D, device=DEV, requires_grad=True)
    theta0 = torch.zeros(1, P, device=DEV)

    def run(share):
        monkeypatch.setattr(ops.GradReducer, "share_packed_grads", share)
        outs = []
        for rep in range(2):
            g = torch.Generator(device="cpu").manual_seed(100 + rep)
            heads = [torch.randn(B, 2 * (D + P), generator=g).to(DEV) for _ in range(2)]
            for h in heads:
                h[:, D + P:] = h[:, D + P:].abs() * 0.5 + 0.1       # (positive scales)
            heads = [h.requires_grad_(True) for h in heads]
            eps = [torch.randn(B, D, generator=g).to(DEV) for _ in range(5)] + [torch.randn(B, P, generator=g).to(DEV) for _ in range(2)]
            _, klj, zj = ops.poe_reparam_kl(theta, heads, [eps[0]], 0, 1 << 2, None, cols=(0, D))
            terms = [klj.sum(), (zj[0] * eps[0]).sum()]
            for m in range(2):
                _, kl, z = ops.poe_reparam_kl(theta, [heads[m]], [eps[1 + 2 * m], eps[2 + 2 * m]], 2, 0b10, None, cols=(0, D))
                _, klp, zp = ops.poe_reparam_kl(theta0, [heads[m]], [eps[5 + m]], 2, 0b10, None, cols=(D, P))
                terms += [kl[1].sum(), (z[0] * z[1]).sum(), klp[1].sum(), (zp[0] ** 2).sum()]
            torch.stack(terms).sum().backward()
            outs.append([h.grad.clone() for h in heads])
        return outs
    a, b = run(False), run(True)
    for rep in range(2):
        for m in range(2):
            check(b[rep][m], a[rep][m].double(), 2e-7, f"d head {m} (pass {rep})")
    assert not ops.GradReducer.packed_grads
