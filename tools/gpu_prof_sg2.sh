#!/bin/bash
# rocprofv3 kernel stats of the StyleGAN2 generator forward+backward (cars 512^2, B=9)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SPEC=${1:-512:9}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sg2 -o sg2 -- python $R/tools/perf_sg2.py $SPEC > $R/gpurun_out/prof_sg2.log 2> $R/gpurun_out/prof_sg2.err
cd $R
rm -f gpurun_out/prof_sg2/*kernel_trace.csv
cat gpurun_out/prof_sg2.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_sg2/sg2_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:40]:
    print('%6.2f%% %5d calls %9.1f us  %s' % (100*float(r['TotalDurationNs'])/tot, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:120]))
PY
