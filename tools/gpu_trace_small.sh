#!/bin/bash
# kernel trace of the inner step at a small local candidate count: busy fraction, gaps,
# launches with < 256 workgroups
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
P2L_ONLY_N=${1:-3} timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_small -o t -- python $R/tools/step_vs_batch.py > $R/gpurun_out/trace_small.log 2>&1
cd $R
grep local gpurun_out/trace_small.log
python tools/underfilled.py $(ls gpurun_out/trace_small/*kernel_trace.csv | head -1) > gpurun_out/underfilled_small.txt 2>&1
python - <<'PY'
import csv, glob, statistics
f = glob.glob('gpurun_out/trace_small/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows))
ev = ev[int(len(ev) * 0.5):]
span = ev[-1][1] - ev[0][0]
busy = 0; cur = ev[0][0]; gaps = []
for s, e in ev:
    if s > cur: gaps.append(s - cur)
    busy += max(0, e - max(s, cur)); cur = max(cur, e)
print('kernels %d span %.2f ms busy %.1f%% gaps n=%d median %.2f us total %.2f ms' % (len(ev), span/1e6, 100*busy/span, len(gaps), statistics.median(gaps)/1e3, sum(gaps)/1e6))
PY
rm -f gpurun_out/trace_small/*kernel_trace.csv
head -40 gpurun_out/underfilled_small.txt
