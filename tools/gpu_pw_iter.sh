#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -5
python tools/bench_pw.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_pw.log
python bench.py --no-cpu-baseline --no-extra --no-fp32-leg --steps 12 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
python - <<'PY'
import json
r = json.load(open('gpurun_out/bench_quick.json'))
print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['conv1x1'])
PY
tail -3 gpurun_out/bench_quick.err
