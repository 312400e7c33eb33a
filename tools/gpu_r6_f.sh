#!/bin/bash
mkdir -p gpurun_out/r6f
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f
cd $R
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q > $O/bench_tests.txt 2>&1
tail -3 $O/bench_tests.txt
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-fp32-leg > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-fp32-leg --eager > $O/bench_eager.json 2> $O/bench_eager.err
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-fp32-leg > $O/bench_graph2.json 2>> $O/bench_graph.err
python - <<PY
import json
for n in ('bench_graph', 'bench_eager', 'bench_graph2'):
    r=json.load(open('$O/%s.json' % n)); ro=r['roofline']
    print(n, r['value'], r['ms_per_step'], r['config']['hip_graph_replay'], ro['frac'], ro['time_share_of_step'], ro['traffic'], ro['traffic_read_write'] and ro['traffic_read_write']['traffic_over_algorithmic'], ro['concurrent']['frac'], ro['concurrent']['time_share_of_step'], ro['concurrent']['traffic'])
PY
tail -3 $O/bench_graph.err
