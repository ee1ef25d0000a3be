mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_hip_ops.py -q -m gpu -k "linear" 2>&1 | tail -5
for B in 512 1000; do python3 bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03b_b$B.json 2>/dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/r03b_b$B.json'));print($B, d['ms_per_step'], d['value'], d['config']['step_flops_fraction_of_f32_mfma_peak'])"; done
MMVAE_GEMM_BIG64=0 python3 bench.py --batch 1000 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('big64 off', d['ms_per_step'])"
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('b128', d['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03b_prof1000 -- python3 bench.py --batch 1000 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find gpurun_out/r03b_prof1000 -name "*kernel_stats.csv" | head -1); head -24 $f | cut -c1-150
