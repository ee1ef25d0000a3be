#!/bin/bash
# same-box comparison of several library builds over several workloads: tools/r04_ab_cfgs.sh "libA.so libB.so" "cfg3 cfg4 ..."
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for cfg in $2; do
  for round in 1 2; do
    for lib in $1; do
      r=$(MMVAE_HIP_LIB=$PWD/$lib python bench.py --config $cfg --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
      echo "$cfg $lib -> $r"
    done
  done
done
