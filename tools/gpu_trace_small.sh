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
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows))
ev = ev[int(len(ev) * 0.5):]
span = ev[-1][1] - ev[0][0]
busy = 0; cur = ev[0][0]; gaps = []
prev = ''
for s, e, n in ev:
    if s > cur: gaps.append((s - cur, prev, n))
    busy += max(0, e - max(s, cur)); cur = max(cur, e); prev = n
gs = [g[0] for g in gaps]
print('kernels %d span %.2f ms busy %.1f%% gaps n=%d median %.2f us total %.2f ms' % (len(ev), span/1e6, 100*busy/span, len(gaps), statistics.median(gs)/1e3, sum(gs)/1e6))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    agg[(a, b)][0] += 1; agg[(a, b)][1] += g
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('%7.1f us in %3d gaps  after %-60s before %s' % (v[1] / 1e3, v[0], k[0], k[1]))
PY
rm -f gpurun_out/trace_small/*kernel_trace.csv
head -12 gpurun_out/underfilled_small.txt
