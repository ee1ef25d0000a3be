#!/bin/bash
# the whole -m gpu suite as the driver runs it, plus the record of every relaxed comparison (profiles/r06_relaxed_bars.json
# is made from gpurun_out/relaxed_bars.jsonl by tools/relaxed_summary.py)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/relaxed_bars.jsonl
python -m pytest tests -q -m gpu "$@" 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
python tools/relaxed_summary.py
