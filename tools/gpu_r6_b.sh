#!/bin/bash
mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b
cd $R
timeout 1500 python -m pytest tests/test_sg2_fullsize_oracle_gpu.py -q -s -m gpu > $O/sg2_fullsize.txt 2>&1
echo "sg2 fullsize rc=$?" >> $O/sg2_fullsize.txt
timeout 600 python -m pytest tests/test_lanes_gpu.py tests/test_bench_gpu.py -x -q > $O/small_tests.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep -v "^$" $O/sg2_fullsize.txt | grep "cars-512\|ffhq-1024\|passed\|failed\|Error" ; tail -3 $O/small_tests.txt
python - <<PY
import json; r=json.load(open('$O/bench.json')); ro=r['roofline']
print(r['value'], r['ms_per_step'], ro['achieved'], ro['frac'], ro['avg_launch_ms'], ro['time_share_of_step'], ro['dominant_kernel'])
print('concurrent', {k: v for k, v in ro.get('concurrent', {}).items() if k not in ('conv1x1', 'what')})
print({k: (v.get('evals_per_s')) for k, v in r['config']['extra'].items()})
print(r['cpu_baseline'])
PY
