"""ResNet-50 image tower: what `encoder: CNN` selects in the reference (models/encoders.py:86-127 -- torchvision's
`resnet50(weights=IMAGENET1K_V1)` -> SiLU -> Linear(1000 -> D) heads).

torchvision is not a dependency: the published ResNet-50 v1.5 topology (He et al. 2016; stride on the 3x3 convolution
of a bottleneck) is laid out here from scratch with torchvision's parameter / buffer names, so a torchvision state_dict
(the ImageNet weights the reference downloads at construction, models/encoders.py:108) loads into it:
    MMVAE_RESNET50_WEIGHTS=/path/to/resnet50.pth   (a `resnet50().state_dict()` file; keys `conv1.weight`, `bn1.*`,
                                                   `layer1.0.conv1.weight`, ..., `fc.bias`)
Without it the tower starts from torchvision's own random initialisation (kaiming-normal fan-out convolutions,
BatchNorm weight 1 / bias 0, default Linear) -- there is no network here to fetch the ImageNet file from.

MI355X mapping: activations are NHWC (rows, C) matrices inside the tower.  The 16 bottlenecks (52 of the 53
convolutions, 52 BatchNorms) run on the fused convolution + BatchNorm engine of csrc/rconv.hip (host side: rconv.py):
implicit GEMM on fp32 MFMA with the im2col tiles staged in LDS, BatchNorm statistics in the GEMM epilogues, normalise +
ReLU in the consumer GEMM's staging, the BatchNorm backward folded into the data- / weight-gradient GEMMs.  k x k
weights are stored channels-last behind their (Cout, Cin, k, k) parameter view (state_dict values are unchanged).  The
stem (7x7 convolution straight from the NCHW image, BatchNorm statistics in its epilogue, max pooling that normalises
while it reads) and the global average pooling ride on the same engine: no im2col / col2im / BatchNorm kernel is left
on the tower's path.
Parity: pinned against an oracle restatement of the same published topology with synthetic weights
(oracle/mmvae_oracle.py: enc_cnn_resnet50), which is pinned against transformers' independent ResNet v1.5
(tests/test_oracle_resnet_independent.py); UNPINNED against torchvision itself, which is absent in this image."""
import math
import os

import torch
import torch.nn as nn

from .. import hipops as H
from .. import ops
from .. import rconv
from .nn_modules import HipLinear

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))      # (planes, blocks, stride of the first block)
EXPANSION = 4


class ConvW(nn.Module):
    """nn.Conv2d(bias=False) parameters, torchvision's initialisation (kaiming_normal_, fan_out, relu); a parameter holder:
    the convolution runs in the fused engine (rconv.py)"""

    def __init__(self, cin, cout, k, stride, pad, channels_last=False):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        if channels_last and k > 1:     # memory (cout, k, k, cin): the fused engine reduces over contiguous channels
            w = w.contiguous(memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)
        self.k, self.stride, self.pad = k, stride, pad


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d parameters / buffers (weight, bias, running_mean, running_var, num_batches_tracked)"""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.momentum, self.eps = 0.1, 1e-5

    def flat_groups(self):
        return [[self.weight, self.bias]]

    def forward(self, x, tap=True):
        """the arithmetic lives in the fused engine (rconv.py); the engine calls this with the BatchNorm's output so that
        forward hooks (the tests' ReLU-mask export) can see it"""
        return x


class Bottleneck(nn.Module):
    """conv1 1x1 -> bn1 -> relu -> conv2 3x3 (stride) -> bn2 -> relu -> conv3 1x1 -> bn3 (+ identity) -> relu.
    Input and output are PRE-activations `s` (the block's consumers apply relu)."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = ConvW(inplanes, planes, 1, 1, 0)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = ConvW(planes, planes, 3, stride, 1, channels_last=True)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = ConvW(planes, planes * EXPANSION, 1, 1, 0)
        self.bn3 = BatchNorm2d(planes * EXPANSION)
        self.downsample = None
        if downsample:
            self.downsample = nn.ModuleList([ConvW(inplanes, planes * EXPANSION, 1, stride, 0),
                                             BatchNorm2d(planes * EXPANSION)])

        self._engine = None

    def engine(self):
        if self._engine is None:
            self._engine = rconv.Block(self)
        return self._engine

    def forward(self, s, B, Hh, W, in_act=H.ACT_RELU):
        """s: (B*H*W, inplanes), consumed through `in_act` (RELU; NONE for the max-pooled stem output)"""
        st = self.conv2.stride
        out = rconv.bottleneck_stack(s, [self.engine()], B, Hh, W, in_act, self.training)
        return out, (Hh - 1) // st + 1, (W - 1) // st + 1


class ResNet50(nn.Module):
    """torchvision.models.resnet50 topology and state_dict names"""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = ConvW(3, 64, 7, 2, 3, channels_last=True)
        self._stem = None
        self.bn1 = BatchNorm2d(64)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(LAYERS):
            layer = []
            for b in range(blocks):
                st = stride if b == 0 else 1
                layer.append(Bottleneck(inplanes, planes, st, downsample=(b == 0)))
                inplanes = planes * EXPANSION
            setattr(self, f"layer{li + 1}", nn.ModuleList(layer))
        self.fc = HipLinear(512 * EXPANSION, num_classes)

    def forward(self, x):
        """x (B,3,H,W) NCHW image batch -> (B, 1000) logits"""
        B, _, Hh, W = x.shape
        if self.training:       # nn.BatchNorm2d counts its training-mode forward passes (one multi-tensor launch)
            torch._foreach_add_([m.num_batches_tracked for m in self.modules() if isinstance(m, BatchNorm2d)], 1)
        if self._stem is None:
            self._stem = rconv.Unit(self.conv1, self.bn1)
        h = rconv.stem(x.float().contiguous(), self._stem, self.training)    # conv1 -> bn1 -> relu -> maxpool
        Hh, W = (Hh + 2 * 3 - 7) // 2 + 1, (W + 2 * 3 - 7) // 2 + 1
        Hh, W = (Hh - 1) // 2 + 1, (W - 1) // 2 + 1
        # the pooled values are already rectified (ACT_NONE); all 16 bottlenecks are one autograd node
        blocks = [blk for li in range(4) for blk in getattr(self, f"layer{li + 1}")]
        # (with the global average pooling of relu(.) behind them: its backward shares a launch with the stack's)
        h = rconv.bottleneck_stack(h, [blk.engine() for blk in blocks], B, Hh, W, H.ACT_NONE, self.training, pool=True)
        return self.fc(h)


def load_torchvision_weights(resnet, path):
    """load a torchvision `resnet50().state_dict()` file (the ImageNet weights the reference fetches itself)"""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = sd.get("state_dict", sd)
    missing, unexpected = resnet.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"{path}: not a torchvision resnet50 state_dict (missing {missing[:3]}, unexpected "
                           f"{unexpected[:3]})")


def maybe_load_pretrained(resnet):
    path = os.environ.get("MMVAE_RESNET50_WEIGHTS")
    if path:
        load_torchvision_weights(resnet, path)
        return True
    return False
