"""Host logic of the fold table (ops.GradReducer._merged / _check_disjoint): no GPU, no library call."""
import pytest

from multimodal_vae_comparison_amd import ops

G = ops.GradReducer


def _spans(segs):
    return sorted({(dp, dp + 4 * ln) for _, dp, _, ln, _ in segs})


def test_continuing_segments_become_one_entry():
    # LayerNorm gamma | beta: columns d..2d of the same partial rows, adjacent destinations
    segs = [(0x1000, 0x9000, 400, 32, 64), (0x1000 + 4 * 32, 0x9000 + 4 * 32, 400, 32, 64)]
    assert G._merged(segs) == [(0x1000, 0x9000, 400, 64, 64)]
    # another row count / stride, a gap in the source or in the destination: nothing to merge
    for other in ((0x1080, 0x9080, 399, 32, 64), (0x1080, 0x9080, 400, 32, 96), (0x1084, 0x9080, 400, 32, 64),
                  (0x1080, 0x9084, 400, 32, 64)):
        assert G._merged([segs[0], other]) == [segs[0], other]


def test_runs_of_different_length_over_one_parameter_are_cut_to_equal_or_disjoint_ranges():
    # a fused text layer registers its six LayerNorm tensors as ONE run of 6 x 16 floats; the launch-per-op form of the same
    # layer (another call of the same module) registers three (gamma | beta) pairs from three other partial buffers
    base, d = 0x20000, 16
    fused = [(0x100000 + 4 * d * k, base + 4 * d * k, 64, d, 6 * d) for k in range(6)]
    pairs = []
    for k in range(3):
        src = 0x200000 + 0x1000 * k
        pairs += [(src, base + 4 * 2 * d * k, 90, d, 2 * d), (src + 4 * d, base + 4 * (2 * d * k + d), 90, d, 2 * d)]
    out = G._merged(fused + pairs)
    assert len(out) == 6 and all(ln == 2 * d for _, _, _, ln, _ in out)
    G._check_disjoint(out)
    assert _spans(out) == [(base + 4 * 2 * d * k, base + 4 * 2 * d * (k + 1)) for k in range(3)]
    # every registered float is still folded exactly once per registration: total (rows x length) is preserved
    assert sum(r * ln for _, _, r, ln, _ in out) == sum(r * ln for _, _, r, ln, _ in fused + pairs)
    # the sources of the cut pieces point at the right columns
    assert sorted(sp for sp, _, r, _, _ in out if r == 64) == [0x100000 + 4 * 2 * d * k for k in range(3)]


def test_partial_overlap_is_refused():
    with pytest.raises(RuntimeError, match="overlap"):
        G._check_disjoint([(0x1000, 0x9000, 4, 32, 32), (0x2000, 0x9000 + 4 * 16, 4, 32, 32)])
    G._check_disjoint([(0x1000, 0x9000, 4, 32, 32), (0x2000, 0x9000, 8, 32, 32), (0x3000, 0x9000 + 4 * 32, 4, 8, 8)])


def test_merging_can_be_switched_off(monkeypatch):
    segs = [(0x1000, 0x9000, 4, 8, 16), (0x1020, 0x9020, 4, 8, 16)]
    monkeypatch.setattr(G, "merge_adjacent", False)
    assert G._merged(segs) == segs
