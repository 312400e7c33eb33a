#!/bin/bash
# GPU busy fraction of the bench step from a rocprofv3 kernel trace (timestamps)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -o g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-leg > /dev/null 2> $R/gpurun_out/gaps.err
cd $R
python tools/underfilled.py $(ls gpurun_out/gaps/*kernel_trace.csv | head -1) > gpurun_out/underfilled.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/gaps/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
# take the last 60% of kernels (steady-state steps)
ev = ev[int(len(ev) * 0.45):]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_end = ev[0][0]; gaps = []
for s, e, n in ev:
    if s > cur_end: gaps.append(s - cur_end)
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
import statistics
print('kernels %d  span %.2f ms  busy %.2f ms (%.1f%%)  gaps: n=%d mean %.2f us median %.2f us, total %.2f ms' % (
    len(ev), span / 1e6, busy / 1e6, 100 * busy / span, len(gaps), statistics.mean(gaps) / 1e3, statistics.median(gaps) / 1e3, sum(gaps) / 1e6))
big = sorted(gaps)[-10:]
print('largest gaps (us):', [round(g / 1e3, 1) for g in big])
PY
rm -f gpurun_out/gaps/*kernel_trace.csv
