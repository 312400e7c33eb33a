#!/bin/bash
# one kernel-iteration round on the GPU: conv kernel tests, per-layer micro-benchmark, bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -5 > gpurun_out/ktests.log
echo "=== kernel tests"; cat gpurun_out/ktests.log
python tools/bench_bf3.py > gpurun_out/bench_bf3.log 2>&1
echo "=== bench_bf3"; cat gpurun_out/bench_bf3.log
python bench.py --no-cpu-baseline --no-extra --no-fp32-leg --steps 12 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "=== bench"; python - <<'PY'
import json
r = json.load(open('gpurun_out/bench_quick.json'))
print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['conv1x1'])
PY
tail -3 gpurun_out/bench_quick.err
