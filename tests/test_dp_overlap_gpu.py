"""MMVAE_DP_OVERLAP=1 (off by default): the data-parallel step cut at the fusion into two graphs -- the
decoders' half of the flat gradients goes on the wire under the encoders' backward.  It must train exactly like the
one-graph step: tools/probe/dp_overlap_check.py runs both forms for 20 steps from the same state (dropout on) and fails
unless the parameters are bit-identical."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_graph_overlap_step_trains_bit_identically():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "dp_overlap_check.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "max |param diff| after 20 steps: 0.0" in r.stdout, r.stdout[-2000:]
