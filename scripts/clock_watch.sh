#!/bin/bash
# Developer: shader clock and socket power while the bench kernel runs (is the MFMA peak of 2.4 GHz reachable under this load?)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python $ROOT/bench.py --steps 60 --warmup 1 --cpu-tiles 0 --no-host-leg --no-other-workloads > /tmp/clock_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python -c "import json; d=json.load(open('/tmp/clock_bench.json')); print('bench', d['value'], d['roofline']['frac'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo " (idle)"
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
