#!/bin/bash
# the round-end gate: every -m gpu test, smoke(), the default bench line
mkdir -p gpurun_out/gate6
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gate6
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt | cut -c1-300
python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json; r=json.load(open('$O/bench.json')); ro=r['roofline']
print(r['value'], r['ms_per_step'], r['config']['hip_graph_replay'], ro['frac'], ro['time_share_of_step'], ro['traffic_read_write']['traffic_over_algorithmic'], ro['concurrent']['frac'])
PY
