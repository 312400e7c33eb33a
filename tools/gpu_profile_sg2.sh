#!/bin/bash
# rocprofv3 kernel stats of the StyleGAN2 configurations (bench.py config.extra: C4 cars 512^2 x 32,
# C5 shard ffhq 1024^2 x 3) + the per-layer conv table of the same process
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in c4 c5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sg2_$c -o r -- python $R/tools/step_sg2_one.py $c $R/gpurun_out/sg2_${c}_layers.txt > $R/gpurun_out/sg2_$c.log 2> $R/gpurun_out/sg2_$c.err
  rm -f $R/gpurun_out/prof_sg2_$c/*kernel_trace.csv
  tail -5 $R/gpurun_out/sg2_$c.log
  head -14 $R/gpurun_out/prof_sg2_$c/*kernel_stats.csv | cut -c1-160
done
