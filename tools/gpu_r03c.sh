mkdir -p gpurun_out; export TMPDIR=/tmp
python3 tools/probe/ffn_time.py
python3 bench.py --config cfg2_rnn --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-400
python3 bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-300
