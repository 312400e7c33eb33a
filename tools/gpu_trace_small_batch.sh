#!/bin/bash
# kernel time of one BigGAN step at a small local candidate count (what one rank of an 8-GPU job runs)
n=${1:-3}
mkdir -p gpurun_out/small_batch
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small_batch
cd /tmp
P2L_ONLY_N=$n timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb_$n -o g -- python $R/tools/step_vs_batch.py > $O/run_$n.txt 2>&1
f=$(ls /tmp/sb_$n/*kernel_trace.csv /tmp/sb_$n/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/step_kernels.py $f > $O/step_kernels_$n.txt 2>&1
tail -2 $O/run_$n.txt; head -60 $O/step_kernels_$n.txt
