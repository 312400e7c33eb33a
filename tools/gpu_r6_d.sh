#!/bin/bash
mkdir -p gpurun_out/r6d
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.txt 2>&1
tail -5 $O/gpu_tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/prof_layers.py > $O/layers.txt 2>/dev/null
python tools/step_vs_batch.py 2>/dev/null | grep "local candidates" > $O/step_vs_batch.txt
python - <<PY
import json; r=json.load(open('$O/bench.json')); ro=r['roofline']
print(r['value'], r['ms_per_step'], ro['achieved'], ro['frac'], ro['avg_launch_ms'], ro['time_share_of_step'], ro['families_ms_per_step'])
print({k: (v.get('evals_per_s')) for k, v in r['config']['extra'].items()})
PY
grep " 64    64 " $O/layers.txt; tail -2 $O/layers.txt; cat $O/step_vs_batch.txt
