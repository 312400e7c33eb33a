#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --lpips-net alex --no-cpu-baseline --no-fp32-leg > gpurun_out/bench_alex.json 2> gpurun_out/bench_alex.err
grep -v cma-es gpurun_out/bench_alex.json; tail -3 gpurun_out/bench_alex.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_alex -o ax -- python $R/bench.py --lpips-net alex --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg > /dev/null 2> $R/gpurun_out/prof_alex.err
cd $R
rm -f gpurun_out/prof_alex/*kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_alex/ax_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:60]:
    n=r['Name']
    if any(k in n for k in ('gconv','maxpool3','conv1_dgrad','lpips','bilinear')) or rows.index(r)<8:
        print('%6.2f%% %5d calls %9.1f us  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, n[:110]))
PY
