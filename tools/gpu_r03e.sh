mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_parity_e2e.py -q -m gpu -k "input_step_inside" 2>&1 | tail -3
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03e_bench.json 2>/dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/r03e_bench.json'));print(d['ms_per_step']);[print(k,v.get('ms_per_step')) for k,v in d['extras'].items() if isinstance(v,dict) and 'ms_per_step' in v]"
