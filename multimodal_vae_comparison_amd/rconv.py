"""Host side of the fused convolution + BatchNorm engine (csrc/rconv.hip) behind the ResNet-50 tower's bottleneck stack.

The reference's `encoder: CNN` is torchvision's resnet50 (models/encoders.py:86-127); its 16 bottlenecks
(conv1x1 -> bn -> relu -> conv3x3 -> bn -> relu -> conv1x1 -> bn (+ shortcut) -> relu) run here as ONE autograd node:
forward and backward are explicit launch sequences in which no BatchNorm, ReLU or im2col pass exists as a kernel of its
own -- normalisation + ReLU ride in the consumer GEMM's LDS staging, the batch statistics in the producer GEMM's
epilogue, and the BatchNorm backward in the data- / weight-gradient GEMMs' staging (see the header of rconv.hip).

Launches per bottleneck: forward 3 convolutions (+1 projection) + 1 shortcut add; backward 3 (+1) data gradients and
3 (+1) weight gradients; the backward statistics of a block's last BatchNorm(s) come out of the NEXT block's first data
gradient (stand-alone kernel only behind the pooling layer)."""
import ctypes
import os

import torch
from torch.autograd import Function

from . import hipops as H
from . import ops

PRE_NONE, PRE_RELU, PRE_BN_RELU = 0, 1, 2
MASK_NONE, MASK_RAW, MASK_BN = 0, 1, 2

_tables = {}


def tables(device, B, Hh, W, K, S, P):
    """(fwd (T, B*Ho*Wo), bwd (T, B*H*W)) int32 source-row tables of a convolution geometry, built once per geometry"""
    key = (device.index, B, Hh, W, K, S, P)
    t = _tables.get(key)
    if t is None:
        Ho, Wo = (Hh + 2 * P - K) // S + 1, (W + 2 * P - K) // S + 1
        fwd = torch.empty(K * K, B * Ho * Wo, dtype=torch.int32, device=device)
        bwd = torch.empty(K * K, B * Hh * W, dtype=torch.int32, device=device)
        ops._call("mmvae_rc_tables", H.ptr(fwd), H.ptr(bwd), B, Hh, W, K, S, P, H.stream())
        t = _tables[key] = (fwd, bwd)
    return t


_row_maps = {}


def parity_row_map(device, B, Hh, W):
    """the B*H*W input pixels of a stride-2 convolution ordered by the parity class (ih % 2, iw % 2) -- classes (0,0), (0,1),
    (1,0), (1,1), raster order inside a class (mmvae_rc_dgrad_t.row_map); None when H or W is odd"""
    if Hh % 2 or W % 2:
        return None
    key = (device.index, B, Hh, W)
    t = _row_maps.get(key)
    if t is None:
        rows = torch.arange(B * Hh * W, dtype=torch.int32).view(B, Hh, W)
        t = torch.cat([rows[:, ph::2, pw::2].reshape(-1) for ph in (0, 1) for pw in (0, 1)]).to(device).contiguous()
        _row_maps[key] = t
    return t


def channels_last_ptr(w):
    """device pointer of a (Cout, Cin, k, k) weight whose memory is (Cout, k, k, Cin)"""
    if w.dim() == 4 and w.shape[2] * w.shape[3] > 1:
        if not w.permute(0, 2, 3, 1).is_contiguous():
            raise RuntimeError("ResNet tower: a k x k convolution weight is not stored channels-last; create the tower "
                               "through models.resnet (and move it with .to(), which preserves the layout)")
    elif not w.is_contiguous():
        raise RuntimeError("ResNet tower: non-contiguous 1x1 convolution weight")
    return H.ptr(w)


BLOCK_DONE_HOOK = None      # callable(rconv.Block): every gradient of that bottleneck has been launched (backward pass)


class Unit:
    """one convolution + the BatchNorm behind it: parameters and the per-step vectors the kernels hand to each other"""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn
        self._buf = {}
        self._gen = {}

    def stamp(self, M):
        """a forward pass over M rows is about to overwrite this unit's per-step vectors (mean / rstd / sc / pqr are kept
        per unit and row count, not per call): returns the generation the matching backward must still find"""
        g = self._gen[M] = self._gen.get(M, 0) + 1
        return g

    def check(self, M, gen):
        """a second forward with the same row count before this backward (two micro-batches, or an eval
        forward in between) has overwritten the statistics and ReLU masks this backward is about to apply"""
        if self._gen.get(M) != gen:
            raise RuntimeError("ResNet tower: the forward pass this backward belongs to is no longer the unit's latest "
                               f"one over {M} rows (generation {gen} vs {self._gen.get(M)}): its BatchNorm statistics were "
                               "overwritten; run backward before the next forward of the same shape")

    def buffers(self, M, device):
        b = self._buf.get(M)
        if b is None:
            C = self.bn.weight.shape[0]
            f = lambda n: torch.empty(n, device=device)
            R = (M + 63) // 64                       # row tiles, + one level-1 pair per 16 of them
            b = self._buf[M] = {"mean": f(C), "rstd": f(C), "sc": f(C), "pqr": f(3 * C),
                                # (+ slack: a stride-2 data gradient walks its rows in 4 parity classes, up to 3 more tiles)
                                "part": f((R + R // 16 + 8) * C * 2), "part_b": f((R + R // 16 + 8) * C * 2),
                                "counter": torch.zeros((C // 64) * (4 + R // 16), dtype=torch.int32, device=device),
                                "tile_tickets": torch.zeros(R * (C // 64), dtype=torch.int32, device=device)}
            w = self.conv.weight
            Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            self._buf.setdefault("tickets", torch.zeros(H.lib().mmvae_rc_wgrad_tickets(Cin, Cout, T), dtype=torch.int32,
                                                        device=device))
        return b

    def conv_ws(self, M, N, K, T, device):
        """split-reduction workspace of a forward / data-gradient GEMM with an (M, N) output"""
        n = H.lib().mmvae_rc_conv_ws_floats(M, N, K, T)
        if n == 0:
            return None
        key = ("cws", M, N)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = torch.empty(n, device=device)
        return t

    def dgrad_tickets(self, rows, device):
        key = ("dt", rows)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = torch.zeros(((rows + 63) // 64 + 4) * (self.conv.weight.shape[1] // 64), dtype=torch.int32,
                                             device=device)
        return t

    def wgrad_ws(self, M, device):
        w = self.conv.weight
        Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
        n = H.lib().mmvae_rc_wgrad_ws_floats(M, Cin, Cout, T)
        key = ("ws", M)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = torch.empty(max(n, 1), device=device)
        return t


KIND_FWD, KIND_DGRAD, KIND_WGRAD = 0, 1, 2


def geom(Hh, W, K, S, P):
    """mmvae_rc_geom_t of a K x K / stride S / padding P convolution over (H, W) maps"""
    return H.RcGeom(Hh, W, (Hh + 2 * P - K) // S + 1, (W + 2 * P - K) // S + 1, K, S, P)


IDENT = (1, 1, 1, 1, 0)     # (H, W, K, S, P) of a row-to-row (1x1, stride 1) job: the kernels skip the pixel arithmetic


def launch(*jobs):
    """independent jobs (H.RcJob) in ONE launch"""
    arr = (H.RcJob * len(jobs))(*jobs)
    ops._call("mmvae_rc_launch", arr, len(jobs), H.stream())


def _p(t):
    return H.ptr(t)


def fwd_job(u, x, M, pre, xb, g, eval_mode):
    """(job, raw output (M, Cout), buffers) of unit u on pre(x); xb = (producer buffers, producer beta) for PRE_BN_RELU;
    g: (H, W, K, S, P) of the input maps"""
    w = u.conv.weight
    Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    b = u.buffers(M, x.device)
    b["gen"] = u.stamp(M)
    y = torch.empty(M, Cout, device=x.device)
    bn = u.bn
    j = H.RcJob()
    j.kind = KIND_FWD
    f = j.f
    f.x, f.w, f.y, f.g = _p(x), channels_last_ptr(w), _p(y), geom(*g)
    if pre == PRE_BN_RELU:
        f.xmean, f.xsc, f.xbeta = _p(xb[0]["mean"]), _p(xb[0]["sc"]), _p(xb[1])
    f.ws, f.tile_ticket = _p(u.conv_ws(M, Cout, Cin, T, x.device)), _p(b["tile_tickets"])
    f.M, f.Cin, f.Cout, f.T, f.pre = M, Cin, Cout, T, pre
    f.gamma, f.beta, f.run_mean, f.run_var = _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var)
    f.mean, f.rstd, f.sc, f.part, f.counter = _p(b["mean"]), _p(b["rstd"]), _p(b["sc"]), _p(b["part"]), _p(b["counter"])
    f.eps, f.momentum, f.eval = float(bn.eps), float(bn.momentum), int(eval_mode)
    return j, y, b


def _fwd(u, x, Min, M, pre, xb, g, eval_mode):
    j, y, b = fwd_job(u, x, M, pre, xb, g, eval_mode)
    launch(j)
    return y, b


def _stat(u, b, Y, eval_mode, grads):
    """mmvae_rc_stat_t of unit u's BatchNorm (raw input Y); grads: {param: (tensor, acc)} filled with dgamma / dbeta"""
    dg, ag = grads[u.bn.weight]
    db, ab = grads[u.bn.bias]
    assert ag == ab
    return H.RcStat(_p(Y), _p(b["mean"]), _p(b["rstd"]), _p(u.bn.weight), _p(b["pqr"]), _p(dg), _p(db), _p(b["part_b"]),
                    _p(b["counter"]), int(ag), int(eval_mode))


def dgrad_job(u, b, G, Y, g, add, add_tbl, mask, mY, mb, out_rows, stats, with_pqr=True, row_map=None):
    w = u.conv.weight
    Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    out = torch.empty(out_rows, Cin, device=G.device)
    j = H.RcJob()
    j.kind = KIND_DGRAD
    d = j.d
    d.G, d.Y, d.pqr, d.w, d.g = _p(G), _p(Y), _p(b["pqr"]) if with_pqr else None, channels_last_ptr(w), geom(*g)
    d.add, d.add_tbl, d.mask, d.mY, d.row_map = _p(add), _p(add_tbl), mask, _p(mY), _p(row_map)
    if mask == MASK_BN:
        d.mmean, d.msc, d.mbeta = _p(mb[0]["mean"]), _p(mb[0]["sc"]), _p(mb[1])
    d.out, d.ws, d.tile_ticket = _p(out), _p(u.conv_ws(out_rows, Cin, Cout, T, G.device)), _p(u.dgrad_tickets(out_rows, G.device))
    d.M, d.Min, d.Cin, d.Cout, d.T, d.nstat = G.shape[0], out_rows, Cin, Cout, T, len(stats)
    for i, st in enumerate(stats):
        d.st[i] = st
    return j, out


def _dgrad(u, b, G, Y, g, add, mask, mY, mb, out_rows, stats, with_pqr=True):
    j, out = dgrad_job(u, b, G, Y, g, add, None, mask, mY, mb, out_rows, stats, with_pqr)
    launch(j)
    return out


def wgrad_job(u, b, G, Y, x, pre, xb, tbl, grads):
    w = u.conv.weight
    Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    M = G.shape[0]
    dw, acc = grads[w]
    j = H.RcJob()
    j.kind = KIND_WGRAD
    g = j.w
    g.G, g.Y, g.pqr, g.x, g.tbl = _p(G), _p(Y), _p(b["pqr"]), _p(x), _p(tbl)
    if pre == PRE_BN_RELU:
        g.xmean, g.xsc, g.xbeta = _p(xb[0]["mean"]), _p(xb[0]["sc"]), _p(xb[1])
    g.dw, g.ws, g.counter = channels_last_ptr(dw), _p(u.wgrad_ws(M, G.device)), _p(u._buf["tickets"])
    g.M, g.Cin, g.Cout, g.T, g.pre, g.accumulate = M, Cin, Cout, T, pre, int(acc)
    return j


def _wgrad(u, b, G, Y, x, pre, xb, tbl, grads):
    launch(wgrad_job(u, b, G, Y, x, pre, xb, tbl, grads))


class Block:
    """a bottleneck's four units and its geometry"""

    def __init__(self, mod):
        self.mod = mod
        self.u1, self.u2, self.u3 = Unit(mod.conv1, mod.bn1), Unit(mod.conv2, mod.bn2), Unit(mod.conv3, mod.bn3)
        self.ud = Unit(mod.downsample[0], mod.downsample[1]) if mod.downsample is not None else None
        self.stride = mod.conv2.stride

    def params(self):
        us = [self.u1, self.u2, self.u3] + ([self.ud] if self.ud else [])
        return [p for u in us for p in (u.conv.weight, u.bn.weight, u.bn.bias)]


def _tap(bn, Y, b):
    """forward hooks on a BatchNorm module (the tests' ReLU-mask export) see bn(Y) in the engine's own arithmetic"""
    if bn._forward_hooks:
        out = torch.empty_like(Y)
        ops._call("mmvae_rc_bn_apply", H.ptr(Y), H.ptr(b["mean"]), H.ptr(b["sc"]), H.ptr(bn.bias), H.ptr(out), Y.shape[0],
                  Y.shape[1], H.stream())
        bn(out, tap=True)


class BottleneckStack(Function):
    """s_out = blocks(s_in): s_in (B*H*W, C) pre-activation consumed through `in_act` by the first block"""

    @staticmethod
    def forward(ctx, s_in, blocks, B, Hh, W, in_act, training, pool, *params):
        """pool: also apply AdaptiveAvgPool2d(1) to relu(s_out) -> (B, C) (resnet50.avgpool; its backward then rides in
        the launch that makes the last BatchNorm's backward statistics)"""
        s = H.f32c(s_in)
        dev = s.device
        saved = []
        act = in_act
        for blk in blocks:
            Min = B * Hh * W
            assert s.shape[0] == Min
            S_ = blk.stride
            Ho, Wo = (Hh - 1) // S_ + 1, (W - 1) // S_ + 1
            M2 = B * Ho * Wo
            pre1 = PRE_RELU if act == H.ACT_RELU else PRE_NONE
            g3, g1 = (Hh, W, 3, S_, 1), (Hh, W, 1, S_, 0)
            ev = not training
            j1, Y1, b1 = fwd_job(blk.u1, s, Min, pre1, None, IDENT, ev)
            Yd = bd = None
            if blk.ud is not None:      # the projection shortcut reads the same input: one launch with conv1
                jd, Yd, bd = fwd_job(blk.ud, s, M2, pre1, None, g1, ev)
                launch(j1, jd)
            else:
                assert S_ == 1
                launch(j1)
            _tap(blk.u1.bn, Y1, b1)
            Y2, b2 = _fwd(blk.u2, Y1, Min, M2, PRE_BN_RELU, (b1, blk.u1.bn.bias), g3, ev)
            _tap(blk.u2.bn, Y2, b2)
            Y3, b3 = _fwd(blk.u3, Y2, M2, M2, PRE_BN_RELU, (b2, blk.u2.bn.bias), IDENT, ev)
            out = torch.empty(M2, Y3.shape[1], device=dev)
            R = Yd if Yd is not None else s
            ops._call("mmvae_rc_blockout", H.ptr(Y3), H.ptr(b3["mean"]), H.ptr(b3["sc"]), H.ptr(blk.u3.bn.bias), H.ptr(R),
                      H.ptr(bd["mean"]) if bd else None, H.ptr(bd["sc"]) if bd else None,
                      H.ptr(blk.ud.bn.bias) if bd else None, int(act == H.ACT_RELU), H.ptr(out), M2, Y3.shape[1], H.stream())
            if blk.u3.bn._forward_hooks:
                blk.u3.bn(out, tap=True)
            saved.append((s, Y1, Y2, Y3, Yd, act, (B, Hh, W, Ho, Wo),
                          (b1["gen"], b2["gen"], b3["gen"], bd["gen"] if bd else None)))
            s, Hh, W, act = out, Ho, Wo, H.ACT_RELU
        ctx.blocks, ctx.saved, ctx.training = blocks, saved, training
        ctx.params = params
        ctx.pool = None
        if pool:
            y = torch.empty(B, s.shape[1], device=dev)
            ops._call("mmvae_avgpool_fwd", H.ptr(s), H.ptr(y), B, Hh * W, s.shape[1], H.ACT_RELU, H.stream())
            ctx.pool = (s, B, Hh * W)
            return y
        return s

    @staticmethod
    def backward(ctx, G):
        blocks, saved, ev = ctx.blocks, ctx.saved, not ctx.training
        G = H.f32c(G)
        dev = G.device
        grads, ret = {}, {}
        for p in ctx.params:
            if p.grad is not None:
                grads[p] = (p.grad, 1)
            else:
                t = torch.empty_like(p)
                grads[p] = (t, 0)
                ret[p] = t
        # Weight gradients ride in the launch of the data gradient that shares their inputs.  (Round 3 also had them queued
        # and sent in batches of up to 8 jobs per launch on a side stream beside the chain of data gradients: measured SLOWER,
        # 4.55 against 4.2 ms/step at batch 24 -- the bulk workgroups take the CU slots the chain's short latency-bound kernels
        # need; giving the chain's stream a higher priority made the graph 3x slower.  Removed in round 4.)
        pending, held = [], []
        nb = H.RC_MAX_JOBS

        def take_wgrads():
            """the queued weight-gradient jobs join the next data-gradient launch"""
            out = pending[:]
            del pending[:]
            return out

        def flush_wgrads(force=False):
            while len(pending) >= nb or (force and pending):
                batch = pending[:nb]
                del pending[:nb]
                launch(*batch)

        ready = False          # the statistics of this block's bn3 (/ projection bn) already came out of the next block
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            s, Y1, Y2, Y3, Yd, act, (B, Hh, W, Ho, Wo), gens = saved[bi]
            Min, M2 = B * Hh * W, B * Ho * Wo
            b1, b2, b3 = blk.u1.buffers(Min, dev), blk.u2.buffers(M2, dev), blk.u3.buffers(M2, dev)
            bd = blk.ud.buffers(M2, dev) if blk.ud else None
            blk.u1.check(Min, gens[0]); blk.u2.check(M2, gens[1]); blk.u3.check(M2, gens[2])
            if blk.ud:
                blk.ud.check(M2, gens[3])
            S_ = blk.stride
            t3 = tables(dev, B, Hh, W, 3, S_, 1)          # the weight gradients' row tables
            t1 = tables(dev, B, Hh, W, 1, S_, 0) if S_ != 1 else (None, None)
            g3 = (Hh, W, 3, S_, 1)
            pre1 = PRE_RELU if act == H.ACT_RELU else PRE_NONE
            if not ready and ctx.pool is not None and not blk.ud:
                s_out, pB, pHW = ctx.pool
                dy, G = G, torch.empty_like(s_out)
                st3 = _stat(blk.u3, b3, Y3, ev, grads)
                ops._call("mmvae_rc_pool_bwd_stats", H.ptr(dy), H.ptr(s_out), H.ptr(G), ctypes.byref(st3), pB, pHW,
                          Y3.shape[1], H.stream())
            elif not ready:
                if ctx.pool is not None:
                    s_out, pB, pHW = ctx.pool
                    dy, G = G, torch.empty_like(s_out)
                    ops._call("mmvae_avgpool_bwd", H.ptr(dy), H.ptr(s_out), H.ptr(G), pB, pHW, s_out.shape[1], H.ACT_RELU,
                              H.stream())
                st3 = _stat(blk.u3, b3, Y3, ev, grads)
                ops._call("mmvae_rc_bn_bwd_stats", H.ptr(G), ctypes.byref(st3), M2, Y3.shape[1], H.stream())
                if blk.ud:
                    std = _stat(blk.ud, bd, Yd, ev, grads)
                    ops._call("mmvae_rc_bn_bwd_stats", H.ptr(G), ctypes.byref(std), M2, Yd.shape[1], H.stream())
            need_in = bi > 0 or ctx.needs_input_grad[0]
            # conv3's data gradient (-> gradient of bn2's output: ReLU mask from Y2, + bn2's statistics) and the projection
            # shortcut's (on ITS output rows; conv1's epilogue adds it through the stride table): both only need G
            j, G2 = dgrad_job(blk.u3, b3, G, Y3, IDENT, None, None, MASK_BN, Y2, (b2, blk.u2.bn.bias), M2,
                              [_stat(blk.u2, b2, Y2, ev, grads)])
            jobs = [j]
            pending.append(wgrad_job(blk.u3, b3, G, Y3, Y2, PRE_BN_RELU, (b2, blk.u2.bn.bias), None, grads))
            add = add_tbl = None
            if blk.ud:
                if need_in:
                    j, add = dgrad_job(blk.ud, bd, G, Yd, IDENT, None, None, MASK_NONE, None, None, M2, [])
                    add_tbl = t1[1][0] if S_ != 1 else None
                    jobs.append(j)
                pending.append(wgrad_job(blk.ud, bd, G, Yd, s, pre1, None, t1[0], grads))
            else:
                add = G
            launch(*jobs, *take_wgrads())
            # conv2 (3x3, stride): data gradient on the rows of the block's input resolution (+ bn1's statistics)
            j, G1 = dgrad_job(blk.u2, b2, G2, Y2, g3, None, None, MASK_BN, Y1, (b1, blk.u1.bn.bias), Min,
                              [_stat(blk.u1, b1, Y1, ev, grads)],
                              row_map=parity_row_map(dev, B, Hh, W) if S_ == 2 else None)
            pending.append(wgrad_job(blk.u2, b2, G2, Y2, Y1, PRE_BN_RELU, (b1, blk.u1.bn.bias), t3[0], grads))
            launch(j, *take_wgrads())
            # conv1: + shortcut, ReLU mask of the block input, and the statistics of the PREVIOUS block's last BatchNorms
            stats = []
            if bi > 0:
                pb = blocks[bi - 1]
                pY3, pYd = saved[bi - 1][3], saved[bi - 1][4]
                stats.append(_stat(pb.u3, pb.u3.buffers(Min, dev), pY3, ev, grads))
                if pb.ud:
                    stats.append(_stat(pb.ud, pb.ud.buffers(Min, dev), pYd, ev, grads))
            held += [G, G1, G2, add]
            pending.append(wgrad_job(blk.u1, b1, G1, Y1, s, pre1, None, None, grads))   # (reads conv1's p, q, r: bn1's
            if need_in:                                                                 # statistics came with conv2's dgrad)
                j, G = dgrad_job(blk.u1, b1, G1, Y1, IDENT, add, add_tbl, MASK_RAW if act == H.ACT_RELU else MASK_NONE, s, None,
                                 Min, stats)
                launch(j, *take_wgrads())
            flush_wgrads()
            ready = True
            # block bi + 1 is final now: its last queued weight gradient went out with this block's first launch
            # (data parallel: parallel.StagedGradReducer sends finished buckets while the rest of the tower runs)
            if BLOCK_DONE_HOOK is not None and bi + 1 < len(blocks):
                BLOCK_DONE_HOOK(blocks[bi + 1])
        flush_wgrads(force=True)
        if BLOCK_DONE_HOOK is not None and blocks:
            BLOCK_DONE_HOOK(blocks[0])
        ctx.saved = None
        return (G if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None) + \
            tuple(ret.get(p) for p in ctx.params)


_stem_tables = {}


def stem_table(device, B, C, Hh, W, K, S, P):
    """(2, B*Ho*Wo) int32: per output pixel the image offset of its window origin and the origin's (h0, w0) packed as the
    stem kernels expect them (mmvae_rc_stem_wgrad)"""
    key = (device.index, B, C, Hh, W, K, S, P)
    t = _stem_tables.get(key)
    if t is None:
        Ho, Wo = (Hh + 2 * P - K) // S + 1, (W + 2 * P - K) // S + 1
        b = torch.arange(B).view(B, 1, 1)
        h0 = (torch.arange(Ho) * S - P).view(1, Ho, 1)
        w0 = (torch.arange(Wo) * S - P).view(1, 1, Wo)
        base = (b * (C * Hh * W) + h0 * W + w0).reshape(-1)
        hw = (((h0 + 0x4000) << 16) | (w0 + 0x4000)).expand(B, Ho, Wo).reshape(-1)
        t = _stem_tables[key] = torch.stack([base, hw]).to(torch.int32).to(device).contiguous()
    return t


class Stem(Function):
    """resnet50.conv1 -> bn1 -> relu -> maxpool on the NCHW image batch: (B, 3, H, W) -> (B*Hp*Wp, 64) pooled (rectified)
    values; one convolution launch (BatchNorm statistics in its epilogue), one pooling launch; backward: the pooling
    backward + bn1's backward statistics in one launch, the weight gradient in another."""

    @staticmethod
    def forward(ctx, x, unit, training, w, gamma, beta):
        x = H.f32c(x)
        dev = x.device
        B, C, Hh, W = x.shape
        K, S, P = unit.conv.k, unit.conv.stride, unit.conv.pad
        g = geom(Hh, W, K, S, P)
        M, Cout = B * g.Ho * g.Wo, w.shape[0]
        b = unit.buffers(M, dev)
        ctx.gen = unit.stamp(M)
        y = torch.empty(M, Cout, device=dev)
        bn = unit.bn
        ops._call("mmvae_rc_stem_fwd", H.ptr(x), channels_last_ptr(w), H.ptr(y), M, C, Cout, K * K, ctypes.byref(g),
                  H.ptr(gamma), H.ptr(beta), H.ptr(bn.running_mean), H.ptr(bn.running_var), H.ptr(b["mean"]), H.ptr(b["rstd"]),
                  H.ptr(b["sc"]), H.ptr(b["part"]), H.ptr(b["counter"]), float(bn.eps), float(bn.momentum), int(not training),
                  H.stream())
        _tap(bn, y, b)
        Hp, Wp = (g.Ho - 1) // 2 + 1, (g.Wo - 1) // 2 + 1
        out = torch.empty(B * Hp * Wp, Cout, device=dev)
        idx = torch.empty(B * Hp * Wp, Cout, device=dev, dtype=torch.int32)
        ops._call("mmvae_rc_maxpool_fwd", H.ptr(y), H.ptr(b["mean"]), H.ptr(b["sc"]), H.ptr(beta), H.ptr(out), H.ptr(idx), B,
                  g.Ho, g.Wo, Cout, H.stream())
        ctx.saved = (x, y, idx, w, gamma, beta)
        ctx.cfg = (unit, training, (B, C, Hh, W, K, S, P))
        return out

    @staticmethod
    def backward(ctx, dy):
        x, y, idx, w, gamma, beta = ctx.saved
        unit, training, (B, C, Hh, W, K, S, P) = ctx.cfg
        dev = x.device
        g = geom(Hh, W, K, S, P)
        M, Cout = y.shape
        b = unit.buffers(M, dev)
        unit.check(M, ctx.gen)
        grads, ret = {}, {}
        for p in (w, gamma, beta):
            if p.grad is not None:
                grads[p] = (p.grad, 1)
            else:
                grads[p] = (torch.empty_like(p), 0)
                ret[p] = grads[p][0]
        G = torch.empty_like(y)
        st = _stat(unit, b, y, not training, grads)
        ops._call("mmvae_rc_maxpool_bwd_stats", H.ptr(H.f32c(dy)), H.ptr(idx), H.ptr(G), H.ptr(b["sc"]), H.ptr(beta),
                  ctypes.byref(st), B, g.Ho, g.Wo, Cout, H.stream())
        lib = H.lib()
        key = ("stem_ws", M)
        ws = unit._buf.get(key)
        if ws is None:
            ws = unit._buf[key] = (torch.empty(lib.mmvae_rc_stem_wgrad_ws_floats(M, C, Cout, K * K), device=dev),
                                   torch.zeros((Cout // 64) * ((K * K * C + 63) // 64), dtype=torch.int32, device=dev))
        dw, acc = grads[w]
        ops._call("mmvae_rc_stem_wgrad", H.ptr(G), H.ptr(y), H.ptr(b["pqr"]), H.ptr(x), H.ptr(stem_table(dev, B, C, Hh, W, K, S, P)),
                  channels_last_ptr(dw), H.ptr(ws[0]), H.ptr(ws[1]), M, C, Cout, K * K, ctypes.byref(g), int(acc), H.stream())
        ctx.saved = None
        return None, None, None, ret.get(w), ret.get(gamma), ret.get(beta)


def stem(x, unit, training):
    return Stem.apply(x, unit, training, unit.conv.weight, unit.bn.weight, unit.bn.bias)


def bottleneck_stack(s_in, blocks, B, Hh, W, in_act, training, pool=False):
    params = [p for blk in blocks for p in blk.params()]
    return BottleneckStack.apply(s_in, blocks, B, Hh, W, in_act, training, pool, *params)
