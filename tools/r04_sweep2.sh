#!/bin/bash
# sweeps on top of a base setting: tools/r04_sweep2.sh "BASE1=a BASE2=b" "KNOB=v" ...
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
base=$1; shift
. "tools/live_knobs.sh"
for kv in $base "$@"; do require_live_knob "${kv%%=*}"; done
run() { env $base $1 python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for round in 1 2; do
  echo "base [$base] -> $(run MMVAE_NOP=1)"
  for kv in "$@"; do echo "  + $kv -> $(run $kv)"; done
done
