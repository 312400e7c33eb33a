#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/diag_grad.py > gpurun_out/diag.log 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
cd $GRAFT_REPO_ROOT
echo "=== diag"; cat gpurun_out/diag.log
echo "=== prof"; cat gpurun_out/prof_bench.json; find gpurun_out/prof | head
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -40 $f; done
