#!/bin/bash
# where the inner step's time goes at the per-rank candidate counts of an 8-GPU job (3 | 2):
# per-layer conv table + rocprofv3 kernel stats of eager steps
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-3}
P2L_GRAPH=0 P2L_POP=$N python tools/prof_layers.py > gpurun_out/layers_pop$N.txt 2>/dev/null
head -50 gpurun_out/layers_pop$N.txt; tail -3 gpurun_out/layers_pop$N.txt
cd /tmp
P2L_GRAPH=0 P2L_ONLY_N=$N timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o s -- python $R/tools/step_vs_batch.py > $R/gpurun_out/small.log 2>&1
cd $R
grep local gpurun_out/small.log
rm -f gpurun_out/prof_small/*kernel_trace.csv
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_small/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
print('kernels %d  total %.2f ms  (12 steps)' % (n, tot / 1e6))
for r in rows[:45]:
    print('%-110s %5s %9.1f us avg %6.2f%%' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
