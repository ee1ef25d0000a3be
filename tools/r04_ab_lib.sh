#!/bin/bash
# same-box A/B of two builds of the library on the cfg2 step: tools/r04_ab_lib.sh libA.so libB.so [batch ...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
A=$1; B=$2; shift 2
for bs in ${@:-128}; do
  for round in 1 2 3; do
    for lib in $A $B; do
      r=$(MMVAE_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --batch $bs --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
      echo "B=$bs $lib -> $r"
    done
  done
done
