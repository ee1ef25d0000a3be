#!/bin/bash
# same-box comparison of several builds of the library on the cfg2 step: tools/r04_ab_multi.sh "libA.so libB.so ..." [batch ...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
LIBS=$1; shift
for bs in ${@:-128}; do
  for round in 1 2 3; do
    for lib in $LIBS; do
      r=$(MMVAE_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --batch $bs --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
      echo "B=$bs $lib -> $r"
    done
  done
done
