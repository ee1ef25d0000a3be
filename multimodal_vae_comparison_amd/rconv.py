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


def channels_last_ptr(w):
    """device pointer of a (Cout, Cin, k, k) weight whose memory is (Cout, k, k, Cin)"""
    if w.dim() == 4 and w.shape[2] * w.shape[3] > 1:
        if not w.permute(0, 2, 3, 1).is_contiguous():
            raise RuntimeError("ResNet tower: a k x k convolution weight is not stored channels-last; create the tower "
                               "through models.resnet (and move it with .to(), which preserves the layout)")
    elif not w.is_contiguous():
        raise RuntimeError("ResNet tower: non-contiguous 1x1 convolution weight")
    return H.ptr(w)


class Unit:
    """one convolution + the BatchNorm behind it: parameters and the per-step vectors the kernels hand to each other"""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn
        self._buf = {}

    def buffers(self, M, device):
        b = self._buf.get(M)
        if b is None:
            C = self.bn.weight.shape[0]
            rt = H.lib().mmvae_rc_row_tile(M, C)
            f = lambda n: torch.empty(n, device=device)
            b = self._buf[M] = {"mean": f(C), "rstd": f(C), "sc": f(C), "pqr": f(3 * C),
                                "part": f(((M + rt - 1) // rt + 1) * C * 2), "part_b": f(((M + 31) // 32 + 1) * C * 2),
                                "counter": torch.zeros(C // 32 + 1, dtype=torch.int32, device=device)}
            w = self.conv.weight
            Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            self._buf.setdefault("tickets", torch.zeros(H.lib().mmvae_rc_wgrad_tickets(Cin, Cout, T), dtype=torch.int32,
                                                        device=device))
        return b

    def wgrad_ws(self, M, device):
        w = self.conv.weight
        Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
        n = H.lib().mmvae_rc_wgrad_ws_floats(M, Cin, Cout, T)
        key = ("ws", M)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = torch.empty(max(n, 1), device=device)
        return t


def _fwd(u, x, Min, M, pre, xb, tbl, eval_mode):
    """raw output (M, Cout) of unit u on pre(x); xb: the producer unit's buffers / beta for PRE_BN_RELU"""
    w = u.conv.weight
    Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    b = u.buffers(M, x.device)
    y = torch.empty(M, Cout, device=x.device)
    xm, xs, xbeta = (H.ptr(xb[0]["mean"]), H.ptr(xb[0]["sc"]), H.ptr(xb[1])) if pre == PRE_BN_RELU else (None, None, None)
    bn = u.bn
    ops._call("mmvae_rc_conv_fwd", H.ptr(x), channels_last_ptr(w), xm, xs, xbeta, H.ptr(tbl), H.ptr(y), M, Cin, Cout, T, pre,
              H.ptr(bn.weight), H.ptr(bn.bias), H.ptr(bn.running_mean), H.ptr(bn.running_var), H.ptr(b["mean"]),
              H.ptr(b["rstd"]), H.ptr(b["sc"]), H.ptr(b["part"]), H.ptr(b["counter"]), float(bn.eps), float(bn.momentum),
              int(eval_mode), H.stream())
    return y, b


def _stat(u, b, Y, eval_mode, grads):
    """mmvae_rc_stat_t of unit u's BatchNorm (raw input Y); grads: {param: (tensor, acc)} filled with dgamma / dbeta"""
    dg, ag = grads[u.bn.weight]
    db, ab = grads[u.bn.bias]
    assert ag == ab
    return H.RcStat(H.ptr(Y), H.ptr(b["mean"]), H.ptr(b["rstd"]), H.ptr(u.bn.weight), H.ptr(b["pqr"]), H.ptr(dg), H.ptr(db),
                    H.ptr(b["part_b"]), H.ptr(b["counter"]), int(ag), int(eval_mode))


def _dgrad(u, b, G, Y, tbl, add, mask, mY, mb, out_rows, stats, with_pqr=True):
    w = u.conv.weight
    Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    M = G.shape[0]
    out = torch.empty(out_rows, Cin, device=G.device)
    mm, ms, mbeta = (H.ptr(mb[0]["mean"]), H.ptr(mb[0]["sc"]), H.ptr(mb[1])) if mask == MASK_BN else (None, None, None)
    st = [ctypes.byref(s) for s in stats] + [None, None]
    ops._call("mmvae_rc_conv_dgrad", H.ptr(G), H.ptr(Y), H.ptr(b["pqr"]) if with_pqr else None, channels_last_ptr(w),
              H.ptr(tbl), H.ptr(add), mask, H.ptr(mY), mm, ms, mbeta, H.ptr(out), M, out_rows, Cin, Cout, T, len(stats),
              st[0], st[1], H.stream())
    return out


def _wgrad(u, b, G, Y, x, pre, xb, tbl, grads):
    w = u.conv.weight
    Cout, Cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    M = G.shape[0]
    dw, acc = grads[w]
    xm, xs, xbeta = (H.ptr(xb[0]["mean"]), H.ptr(xb[0]["sc"]), H.ptr(xb[1])) if pre == PRE_BN_RELU else (None, None, None)
    ops._call("mmvae_rc_conv_wgrad", H.ptr(G), H.ptr(Y), H.ptr(b["pqr"]), H.ptr(x), xm, xs, xbeta, H.ptr(tbl),
              channels_last_ptr(dw), H.ptr(u.wgrad_ws(M, G.device)), H.ptr(u._buf["tickets"]), M, Cin, Cout, T, pre,
              int(acc), H.stream())


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
    def forward(ctx, s_in, blocks, B, Hh, W, in_act, training, *params):
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
            t3 = tables(dev, B, Hh, W, 3, S_, 1)
            t1 = tables(dev, B, Hh, W, 1, S_, 0) if S_ != 1 else (None, None)
            ev = not training
            Y1, b1 = _fwd(blk.u1, s, Min, Min, pre1, None, None, ev)
            _tap(blk.u1.bn, Y1, b1)
            Y2, b2 = _fwd(blk.u2, Y1, Min, M2, PRE_BN_RELU, (b1, blk.u1.bn.bias), t3[0], ev)
            _tap(blk.u2.bn, Y2, b2)
            Y3, b3 = _fwd(blk.u3, Y2, M2, M2, PRE_BN_RELU, (b2, blk.u2.bn.bias), None, ev)
            Yd = bd = None
            if blk.ud is not None:
                Yd, bd = _fwd(blk.ud, s, Min, M2, pre1, None, t1[0], ev)
            else:
                assert S_ == 1
            out = torch.empty(M2, Y3.shape[1], device=dev)
            R = Yd if Yd is not None else s
            ops._call("mmvae_rc_blockout", H.ptr(Y3), H.ptr(b3["mean"]), H.ptr(b3["sc"]), H.ptr(blk.u3.bn.bias), H.ptr(R),
                      H.ptr(bd["mean"]) if bd else None, H.ptr(bd["sc"]) if bd else None,
                      H.ptr(blk.ud.bn.bias) if bd else None, int(act == H.ACT_RELU), H.ptr(out), M2, Y3.shape[1], H.stream())
            if blk.u3.bn._forward_hooks:
                blk.u3.bn(out, tap=True)
            saved.append((s, Y1, Y2, Y3, Yd, act, (B, Hh, W, Ho, Wo)))
            s, Hh, W, act = out, Ho, Wo, H.ACT_RELU
        ctx.blocks, ctx.saved, ctx.training = blocks, saved, training
        ctx.params = params
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
        ready = False          # the statistics of this block's bn3 (/ projection bn) already came out of the next block
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            s, Y1, Y2, Y3, Yd, act, (B, Hh, W, Ho, Wo) = saved[bi]
            Min, M2 = B * Hh * W, B * Ho * Wo
            b1, b2, b3 = blk.u1.buffers(Min, dev), blk.u2.buffers(M2, dev), blk.u3.buffers(M2, dev)
            bd = blk.ud.buffers(M2, dev) if blk.ud else None
            S_ = blk.stride
            t3 = tables(dev, B, Hh, W, 3, S_, 1)
            t1 = tables(dev, B, Hh, W, 1, S_, 0) if S_ != 1 else (None, None)
            pre1 = PRE_RELU if act == H.ACT_RELU else PRE_NONE
            if not ready:
                st3 = _stat(blk.u3, b3, Y3, ev, grads)
                ops._call("mmvae_rc_bn_bwd_stats", H.ptr(G), ctypes.byref(st3), M2, Y3.shape[1], H.stream())
                if blk.ud:
                    std = _stat(blk.ud, bd, Yd, ev, grads)
                    ops._call("mmvae_rc_bn_bwd_stats", H.ptr(G), ctypes.byref(std), M2, Yd.shape[1], H.stream())
            # conv3: data gradient -> gradient of bn2's output (ReLU mask from Y2) + bn2's statistics; weight gradient
            G2 = _dgrad(blk.u3, b3, G, Y3, None, None, MASK_BN, Y2, (b2, blk.u2.bn.bias), M2, [_stat(blk.u2, b2, Y2, ev, grads)])
            _wgrad(blk.u3, b3, G, Y3, Y2, PRE_BN_RELU, (b2, blk.u2.bn.bias), None, grads)
            # conv2 (3x3, stride): rows of the block's input resolution
            G1 = _dgrad(blk.u2, b2, G2, Y2, t3[1], None, MASK_BN, Y1, (b1, blk.u1.bn.bias), Min,
                        [_stat(blk.u1, b1, Y1, ev, grads)])
            _wgrad(blk.u2, b2, G2, Y2, Y1, PRE_BN_RELU, (b1, blk.u1.bn.bias), t3[0], grads)
            # shortcut
            if blk.ud:
                add = _dgrad(blk.ud, bd, G, Yd, t1[1], None, MASK_NONE, None, None, Min, [])
                _wgrad(blk.ud, bd, G, Yd, s, pre1, None, t1[0], grads)
            else:
                add = G
            # conv1: + shortcut, ReLU mask of the block input, and the statistics of the PREVIOUS block's last BatchNorms
            stats = []
            if bi > 0:
                pb = blocks[bi - 1]
                pY3, pYd = saved[bi - 1][3], saved[bi - 1][4]
                stats.append(_stat(pb.u3, pb.u3.buffers(Min, dev), pY3, ev, grads))
                if pb.ud:
                    stats.append(_stat(pb.ud, pb.ud.buffers(Min, dev), pYd, ev, grads))
            need_in = bi > 0 or ctx.needs_input_grad[0]
            if need_in:
                G = _dgrad(blk.u1, b1, G1, Y1, None, add, MASK_RAW if act == H.ACT_RELU else MASK_NONE, s, None, Min, stats)
            _wgrad(blk.u1, b1, G1, Y1, s, pre1, None, None, grads)
            ready = True
        ctx.saved = None
        return (G if ctx.needs_input_grad[0] else None, None, None, None, None, None, None) + \
            tuple(ret.get(p) for p in ctx.params)


def bottleneck_stack(s_in, blocks, B, Hh, W, in_act, training):
    params = [p for blk in blocks for p in blk.params()]
    return BottleneckStack.apply(s_in, blocks, B, Hh, W, in_act, training, *params)
