#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40 > gpurun_out/tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
cd $GRAFT_REPO_ROOT
echo "=== tests"; cat gpurun_out/tests.log
echo "=== bench"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
echo "=== prof"; cat gpurun_out/prof_bench.json; tail -3 gpurun_out/prof.err; find gpurun_out/prof -name "*stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -30 $f; done
