"""Host-side pieces of the ResNet engine (rconv.py) that need no GPU: the parity-class row order of stride-2 data
gradients and the stem's window-origin table against their definitions; the channels-last parameter views of flat.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_vae_comparison_amd import rconv        # noqa: E402
from multimodal_vae_comparison_amd.flat import FlatParams  # noqa: E402
from multimodal_vae_comparison_amd.models.resnet import Bottleneck  # noqa: E402

CPU = torch.device("cpu")


def test_parity_row_map_orders_pixels_by_class():
    B, Hh, W = 3, 6, 4
    m = rconv.parity_row_map(CPU, B, Hh, W)
    assert sorted(m.tolist()) == list(range(B * Hh * W))           # a permutation
    n = B * Hh * W // 4
    for cls, (ph, pw) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        rows = m[cls * n:(cls + 1) * n]
        ih, iw = (rows // W) % Hh, rows % W
        assert bool(((ih % 2) == ph).all()) and bool(((iw % 2) == pw).all())
        assert rows.tolist() == sorted(rows.tolist())               # raster order inside a class
    assert rconv.parity_row_map(CPU, 2, 7, 4) is None              # odd maps: raster order, all taps


def test_stem_table_is_the_window_origin():
    B, C, Hh, W, K, S, P = 2, 3, 10, 12, 7, 2, 3
    t = rconv.stem_table(CPU, B, C, Hh, W, K, S, P)
    Ho, Wo = (Hh + 2 * P - K) // S + 1, (W + 2 * P - K) // S + 1
    assert tuple(t.shape) == (2, B * Ho * Wo)
    for b in range(B):
        for oh in range(Ho):
            for ow in range(Wo):
                r = (b * Ho + oh) * Wo + ow
                h0, w0 = oh * S - P, ow * S - P
                assert int(t[0, r]) == b * C * Hh * W + h0 * W + w0
                assert int(t[1, r]) == ((h0 + 0x4000) << 16) | (w0 + 0x4000)


def test_flat_buffers_hold_kxk_weights_channels_last():
    torch.manual_seed(0)
    blk = Bottleneck(64, 64, 2, True)
    w0 = blk.conv2.weight.detach().clone()
    flat = FlatParams(blk)
    w = blk.conv2.weight
    o = flat.offset_of[id(w)]
    assert torch.equal(w.detach(), w0) and tuple(w.stride()) == (576, 1, 192, 64)
    assert torch.equal(flat.data[o:o + w.numel()].view(64, 3, 3, 64), w0.permute(0, 2, 3, 1))
    assert w.grad.data_ptr() == flat.grad.data_ptr() + 4 * o and tuple(w.grad.stride()) == tuple(w.stride())
    w1 = blk.conv1.weight                                   # 1x1: plain contiguous
    assert w1.is_contiguous() and torch.equal(FlatParams.flatten_like(w1.detach(), w1), w1.detach().reshape(-1))


def test_unit_generation_guard():
    """a unit's per-step vectors live per (unit, row count); a backward whose forward is no longer the latest
    one over that row count must refuse instead of applying another pass's statistics"""
    import pytest
    torch.manual_seed(0)
    blk = Bottleneck(64, 64, 1, True)
    u = rconv.Unit(blk.conv1, blk.bn1)
    g1 = u.stamp(128)
    u.check(128, g1)
    g_other = u.stamp(64)            # another row count does not disturb it
    u.check(128, g1)
    u.check(64, g_other)
    g2 = u.stamp(128)                # a second forward over the same rows does
    assert g2 == g1 + 1
    u.check(128, g2)
    with pytest.raises(RuntimeError, match="no longer the unit's latest"):
        u.check(128, g1)


def test_resnet50_weights_file_is_loaded(tmp_path, monkeypatch):
    """MMVAE_RESNET50_WEIGHTS: a torchvision-layout `resnet50().state_dict()` file (the ImageNet weights the reference
    downloads at construction, models/encoders.py:108) initialises the tower; anything else is refused"""
    import torch
    from multimodal_vae_comparison_amd.models import resnet
    torch.manual_seed(3)
    src = resnet.ResNet50()
    path = tmp_path / "resnet50.pth"
    torch.save(src.state_dict(), path)
    torch.manual_seed(4)
    dst = resnet.ResNet50()
    assert not torch.equal(dst.layer3[2].conv2.weight, src.layer3[2].conv2.weight)
    monkeypatch.delenv("MMVAE_RESNET50_WEIGHTS", raising=False)
    assert resnet.maybe_load_pretrained(dst) is False
    monkeypatch.setenv("MMVAE_RESNET50_WEIGHTS", str(path))
    assert resnet.maybe_load_pretrained(dst) is True
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    bad = tmp_path / "bad.pth"
    torch.save({"conv1.weight": torch.zeros(1)}, bad)
    monkeypatch.setenv("MMVAE_RESNET50_WEIGHTS", str(bad))
    with pytest.raises(RuntimeError):
        resnet.maybe_load_pretrained(resnet.ResNet50())


def test_staged_ranges_only_for_one_contiguous_tower():
    """parallel.resnet_block_ranges: block ranges are handed to the staged reducer only when they tile ONE
    contiguous region of the flat buffer; two towers (a gap holding the other tower's layers) get one collective."""
    import torch.nn as nn
    from multimodal_vae_comparison_amd import parallel
    assert parallel.ranges_tile_one_region([(0, 10), (10, 20), (22, 30)])
    assert not parallel.ranges_tile_one_region([(0, 10), (14, 20)])          # a gap wider than the group alignment
    assert not parallel.ranges_tile_one_region([(10, 20), (0, 10)])          # not in module order
    assert not parallel.ranges_tile_one_region([])
    torch.manual_seed(0)
    one = nn.Sequential(Bottleneck(64, 16, 1, True), Bottleneck(64, 16, 1, False))
    flat = FlatParams(one)
    ranges, mods = parallel.resnet_block_ranges(one, flat)
    assert len(ranges) == 2 and mods[0] is one[1] and ranges[0][0] >= ranges[1][1]     # last block first

    class Two(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Sequential(Bottleneck(64, 16, 1, True), Bottleneck(64, 16, 1, False))
            self.mid = nn.Linear(40, 40)          # another tower's head / stem in between
            self.b = nn.Sequential(Bottleneck(64, 16, 1, True), Bottleneck(64, 16, 1, False))
    two = Two()
    flat2 = FlatParams(two)
    assert parallel.resnet_block_ranges(two, flat2) == ([], [])
