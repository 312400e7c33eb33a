#!/bin/bash
# gpurun_out/r6 (tools/gpu_round6.sh) -> profiles/round6_* : the names DESIGN.md / bench.py cite
set -e
S=gpurun_out/r6; D=profiles
cp $S/bench.json $D/round6_bench.json
cp $S/layers.txt $D/round6_layers.txt
cp $S/conv1x1_roofline.txt $D/round6_conv1x1_roofline.txt
cp $S/step_vs_batch.txt $D/round6_step_vs_batch.txt
cp $S/traffic.json $D/round6_traffic.json
for p in "eb18 one_pass_of_18" "eb9 chunks_of_9"; do
  set -- $p
  cp $S/prof_$1/r_kernel_stats.csv $D/round6_kernel_stats_$2.csv
  cp $S/step_kernels_$1.txt $D/round6_step_kernels_$2.txt
  cp $S/pmc_fetch_summary_$1.csv $D/round6_pmc_fetch_summary_$2.csv
  cp $S/pmc_write_summary_$1.csv $D/round6_pmc_write_summary_$2.csv
  cp $S/prof_bench_$1.json $D/round6_bench_under_rocprof_$2.json
done
cp $S/prof_sg2_c4/r_kernel_stats.csv $D/round6_sg2_512_kernel_stats.csv
cp $S/prof_sg2_c5/r_kernel_stats.csv $D/round6_sg2_1024_kernel_stats.csv
cp $S/sg2_c4_layers.txt $D/round6_sg2_512_layers.txt
cp $S/sg2_c5_layers.txt $D/round6_sg2_1024_layers.txt
echo copied
