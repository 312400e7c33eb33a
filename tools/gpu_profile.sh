#!/bin/bash
# rocprofv3 kernel stats + HBM PMC counters (separate passes) of the bench command
# usage: gpurun -- 'bash tools/gpu_profile.sh <commit id>'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
cd $R
ls gpurun_out/pmc_fetch gpurun_out/pmc_write
python tools/pmc_summary.py $(ls gpurun_out/pmc_fetch/*counter_collection.csv | head -1) gpurun_out/pmc_fetch_summary.csv
python tools/pmc_summary.py $(ls gpurun_out/pmc_write/*counter_collection.csv | head -1) gpurun_out/pmc_write_summary.csv
python tools/traffic_json.py gpurun_out/pmc_fetch_summary.csv gpurun_out/pmc_write_summary.csv gpurun_out/traffic.json "${1:-unknown}" "$(hostname)" "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra"
rm -f gpurun_out/pmc_fetch/*counter_collection.csv gpurun_out/pmc_write/*counter_collection.csv gpurun_out/prof/*kernel_trace.csv
echo "=== stats"; head -25 gpurun_out/prof/r1_kernel_stats.csv
cat gpurun_out/prof_bench.json
