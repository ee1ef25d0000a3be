#!/bin/bash
# same-box A/B of one env knob on a bench workload: tools/ab_knob.sh KNOB a b [batch ...]; interleaved pairs, 3 rounds
# (CFG=cfg2 by default; the knob must be one the sources still read: tools/live_knobs.sh)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
knob=$1; a=$2; b=$3; shift 3
. tools/live_knobs.sh
require_live_knob "$knob"
batches=${@:-128}
for bs in $batches; do
  for round in 1 2 3; do
    for v in $a $b; do
      r=$(env $knob=$v python bench.py --config ${CFG:-cfg2} --no-cpu-baseline --no-extras --batch $bs --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], int(d["value"]), d["config"].get("abi_calls_per_step"))')
      echo "B=$bs $knob=$v -> $r"
    done
  done
done
