#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R
rm -f gpurun_out/prof/*kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof/r1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:32]:
    print('%6.2f%% %5d calls %9.1f us  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:110]))
PY
