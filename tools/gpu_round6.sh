#!/bin/bash
# The round-6 measurement set, one gpurun call at one commit:
#   usage: gpurun --timeout 2400 -- 'bash tools/gpu_round6.sh <commit>'
# bench line (all legs); rocprofv3 kernel tables of the TIMED configuration (reference chunks of 9: the tracer
# serialises the two lanes, so these are 9-candidate launches alone) and of the one-pass-of-18 configuration the
# top-level roofline record is measured at; HBM PMC passes at BOTH execution batches; per-layer tables; 1x1
# roofline table; step vs local batch; StyleGAN2 C4 / C5 kernel tables.
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python tools/prof_layers.py > $O/layers.txt 2>/dev/null
python tools/roofline_1x1.py $O/layers.txt > $O/conv1x1_roofline.txt
python tools/step_vs_batch.py 2>/dev/null | grep "local candidates" > $O/step_vs_batch.txt
cd /tmp
for eb in 9 18; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eb$eb -o r -- python $R/bench.py --pmc-run --steps 3 --warmup 1 --exec-batch $eb > $O/prof_bench_eb$eb.json 2> $O/prof_eb$eb.err
  python $R/tools/step_kernels.py $(ls $O/prof_eb$eb/*kernel_trace.csv | head -1) > $O/step_kernels_eb$eb.txt 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_eb$eb -o f -- python $R/bench.py --pmc-run --steps 2 --warmup 1 --exec-batch $eb > $O/pmc_fetch_eb$eb.json 2> $O/pmc_fetch_eb$eb.err
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_eb$eb -o w -- python $R/bench.py --pmc-run --steps 2 --warmup 1 --exec-batch $eb > $O/pmc_write_eb$eb.json 2> $O/pmc_write_eb$eb.err
  python $R/tools/pmc_summary.py $(ls $O/pmc_fetch_eb$eb/*counter_collection.csv | head -1) $O/pmc_fetch_summary_eb$eb.csv > /dev/null
  python $R/tools/pmc_summary.py $(ls $O/pmc_write_eb$eb/*counter_collection.csv | head -1) $O/pmc_write_summary_eb$eb.csv > /dev/null
done
python $R/tools/traffic_json.py $O/traffic.json "${1:-unknown}" "$(hostname)" \
  $O/pmc_fetch_summary_eb18.csv $O/pmc_write_summary_eb18.csv $O/pmc_fetch_eb18.json \
  $O/pmc_fetch_summary_eb9.csv $O/pmc_write_summary_eb9.csv $O/pmc_fetch_eb9.json > /dev/null
for c in c4 c5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sg2_$c -o r -- python $R/tools/step_sg2_one.py $c $O/sg2_${c}_layers.txt > $O/sg2_$c.log 2> $O/sg2_$c.err
done
rm -f $O/pmc_*/*counter_collection.csv $O/prof_*/*kernel_trace.csv
cd $R
python - <<PY
import json; r=json.load(open('$O/bench.json')); ro=r['roofline']
print(r['value'], r['ms_per_step'], ro['achieved'], ro['frac'], ro['avg_launch_ms'], ro['time_share_of_step'], ro['dominant_kernel'])
print('traffic', ro['traffic'], ro['traffic_read_write'], ro['traffic_source'])
print('concurrent', {k: v for k, v in ro.get('concurrent', {}).items() if k not in ('conv1x1', 'what')})
print({k: (v.get('evals_per_s')) for k, v in r['config']['extra'].items()})
print(r['cpu_baseline'])
PY
cat $O/step_vs_batch.txt; tail -3 $O/conv1x1_roofline.txt; head -8 $O/step_kernels_eb18.txt; head -8 $O/step_kernels_eb9.txt; cat $O/traffic.json | head -40
