#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
P2L_ONLY_N=${1:-2} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o s -- python $R/tools/step_vs_batch.py > $R/gpurun_out/prof_small.log 2>&1
cd $R
rm -f gpurun_out/prof_small/*kernel_trace.csv
grep local gpurun_out/prof_small.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_small/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total kernel ms', tot/1e6, 'launches', calls, '-> per step (11 steps):', tot/1e6/11, 'ms,', calls/11, 'launches')
for r in rows[:14]:
    print('%6.2f%% %5d calls %9.1f us  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
