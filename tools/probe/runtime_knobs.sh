#!/bin/bash
run() { r=$(env "$@" python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>/dev/null); echo "$* -> $r"; }
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run GPU_STREAMOPS_CP_WAIT=0
run GPU_STREAMOPS_CP_WAIT=1
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run AMD_DIRECT_DISPATCH=0
run X=0
