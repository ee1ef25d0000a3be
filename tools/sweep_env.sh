#!/bin/bash
# bench.py under every combination of the given env knobs: tools/sweep_env.sh "A=0,1" "B=x,y" ...
combos=("")
. "$(dirname "$0")/live_knobs.sh"
for spec in "$@"; do require_live_knob "${spec%%=*}"; done
for spec in "$@"; do
  name=${spec%%=*}; vals=${spec#*=}
  next=()
  for c in "${combos[@]}"; do for v in ${vals//,/ }; do next+=("$c $name=$v"); done; done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  r=$(env $c python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')
  echo "$c -> $r"
done
