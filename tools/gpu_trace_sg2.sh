#!/bin/bash
# kernel trace of ONE step of a StyleGAN2 configuration (tools/step_sg2_one.py c5|c4)
cfg=${1:-c5}
mkdir -p gpurun_out/trace_sg2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$cfg -o r -- python $R/tools/step_sg2_one.py $cfg > $R/gpurun_out/trace_sg2/$cfg.log 2>&1
f=$(ls /tmp/tr_$cfg/*kernel_trace.csv /tmp/tr_$cfg/*/*kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_last_step.py $f $R/gpurun_out/trace_sg2/${cfg}_last_step.txt | head -70
