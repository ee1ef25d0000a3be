#!/bin/bash
# same-box A/B of one env knob on the cfg2 step: tools/r04_ab.sh KNOB a b [batch ...]; interleaved pairs, 3 rounds
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
knob=$1; a=$2; b=$3; shift 3
. tools/live_knobs.sh
require_live_knob "$knob"
batches=${@:-128}
for bs in $batches; do
  for round in 1 2 3; do
    for v in $a $b; do
      r=$(env $knob=$v python bench.py --no-cpu-baseline --no-extras --batch $bs --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], int(d["value"]), d["config"].get("abi_calls_per_step"))')
      echo "B=$bs $knob=$v -> $r"
    done
  done
done
