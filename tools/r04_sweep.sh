#!/bin/bash
# one-knob sweeps of the cfg2 step against the default, interleaved (2 rounds): tools/r04_sweep.sh "KNOB=v" ...
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
. "tools/live_knobs.sh"
for kv in "$@"; do require_live_knob "${kv%%=*}"; done
run() { env $1 python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for round in 1 2; do
  echo "default -> $(run MMVAE_NOP=1)"
  for kv in "$@"; do echo "$kv -> $(run $kv)"; done
done
