#!/bin/bash
# same-box A/B of two builds of the library on the cfg2 step: tools/probe/ab_libs.sh libA.so libB.so "batches" [rounds]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
A=$1; B=$2; BS=${3:-128}; R=${4:-3}
for b in $BS; do for r in $(seq 1 $R); do for lib in $A $B; do
  v=$(MMVAE_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --batch $b --steps $([ $b -le 256 ] && echo 300 || echo 60) --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
  echo "B=$b $lib $v"
done; done; done
