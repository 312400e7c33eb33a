#!/bin/bash
# socket power / clocks while the bench step runs (is the conv kernel clock- or power-limited?)
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 500 --warmup 2 --no-cpu-baseline --no-fp32-leg > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 10
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
python -c "import json;d=json.loads(open('gpurun_out/power_bench.json').read().strip().splitlines()[-1]);print('evals/s',d['value'],'conv TF',d['roofline']['achieved'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i power
