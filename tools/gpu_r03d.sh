mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_parity_e2e.py -q -m gpu -k "input_step_inside or load_batch_compact or dropout or graph_replay or captured_step" 2>&1 | tail -8
python -m pytest tests/test_hip_ops.py tests/test_curve_parity.py -q -m gpu -k "dropout or ffn or curve or attention or layernorm or txt_layer" 2>&1 | tail -5
python3 tools/probe/ffn_time.py
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03d_bench.json 2>/dev/null; python3 -c "
import json;d=json.load(open('gpurun_out/r03d_bench.json'));print(d['ms_per_step']);[print(k,v.get('ms_per_step')) for k,v in d['extras'].items() if isinstance(v,dict) and 'ms_per_step' in v]"
