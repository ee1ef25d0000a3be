#!/bin/bash
# bench.py at several per-GPU batch sizes (robustness + where the step stops being latency-bound)
for b in "$@"; do
  python bench.py --no-cpu-baseline --no-extras --batch $b --steps 100 --warmup 10 2>/dev/null | tail -1 | B=$b python -c 'import sys,json,os; d=json.loads(sys.stdin.read()); print("B", os.environ["B"], "ms/step", d["ms_per_step"], "samples/s", int(d["value"]), "conv2 frac", d["roofline"]["frac"])'
done
