#!/bin/bash
# The round-5 measurement set, one gpurun call at one commit:
#   usage: gpurun --timeout 2400 -- 'bash tools/gpu_round5.sh <commit>'
# bench line (all legs), rocprofv3 kernel stats + HBM PMC passes of the bench command, per-layer tables
# (18 and 2 candidates), 1x1 roofline table, step vs local batch, per-launch micro benchmark of the 3x3
# kernel forms, small-batch kernel stats
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python tools/prof_layers.py > $O/layers.txt 2>/dev/null
python tools/roofline_1x1.py $O/layers.txt > $O/conv1x1_roofline.txt
python tools/step_vs_batch.py 2>/dev/null | grep "local candidates" > $O/step_vs_batch.txt
python tools/bench_h2.py 2>/dev/null | grep "^18 x" > $O/bench_h2.txt
python tools/bench_wino_blocks.py 2>/dev/null | grep -v amdgpu > $O/wino_blocks.txt
python tools/bench_pw_h2.py 2>/dev/null | grep -v amdgpu > $O/pw_h2.txt
python tools/splitk_sweep.py 2>/dev/null | grep -v amdgpu > $O/splitk_sweep.txt
REPS=40 python tools/ulp_hunt.py 2>/dev/null | tail -2 > $O/ulp_hunt.txt
python tools/bwd_batch_bits.py 2>/dev/null | grep -v "amdgpu\|cma-es" > $O/bwd_batch_bits.txt
P2L_GRAPH=0 P2L_POP=2 python tools/prof_layers.py > $O/layers_pop2.txt 2>/dev/null
P2L_GRAPH=0 P2L_POP=3 python tools/prof_layers.py > $O/layers_pop3.txt 2>/dev/null
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $O/prof_bench.json 2> $O/prof.err
P2L_GRAPH=0 P2L_ONLY_N=2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -o s -- python $R/tools/step_vs_batch.py > $O/small.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
python tools/pmc_summary.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) $O/pmc_fetch_summary.csv > /dev/null
python tools/pmc_summary.py $(ls $O/pmc_write/*counter_collection.csv | head -1) $O/pmc_write_summary.csv > /dev/null
python tools/traffic_json.py $O/pmc_fetch_summary.csv $O/pmc_write_summary.csv $O/traffic.json "${1:-unknown}" "$(hostname)" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra" > /dev/null
rm -f $O/pmc_fetch/*counter_collection.csv $O/pmc_write/*counter_collection.csv $O/prof/*kernel_trace.csv $O/prof_small/*kernel_trace.csv
python - <<PY
import json; r=json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'], r['roofline']['algorithmic'], r['roofline']['conv1x1']['achieved'], r['telemetry'], r['cpu_baseline']['value'])
print({k: (v.get('evals_per_s')) for k, v in r['config']['extra'].items()})
PY
cat $O/step_vs_batch.txt; tail -3 $O/conv1x1_roofline.txt; head -14 $O/prof/r1_kernel_stats.csv | cut -c1-150; cat $O/traffic.json | head -30
