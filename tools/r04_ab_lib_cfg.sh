#!/bin/bash
# same-box A/B of two builds of the library on another workload: tools/r04_ab_lib_cfg.sh libA.so libB.so cfg [batch]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
A=$1; B=$2; cfg=$3; bs=${4:-}
for round in 1 2 3; do
  for lib in $A $B; do
    r=$(MMVAE_HIP_LIB=$PWD/$lib python bench.py --config $cfg ${bs:+--batch $bs} --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
    echo "$cfg ${bs:+B=$bs} $lib -> $r"
  done
done
