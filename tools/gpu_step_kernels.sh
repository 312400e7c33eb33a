#!/bin/bash
# kernel time of one step by name: gpurun -- 'bash tools/gpu_step_kernels.sh'
mkdir -p gpurun_out/stepk
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stepk
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o g -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $O/bench.json 2> $O/err.log
python $R/tools/step_kernels.py $(ls $O/prof/*kernel_trace.csv | head -1) > $O/step_kernels.txt 2>&1
rm -f $O/prof/*kernel_trace.csv
cat $O/step_kernels.txt
