#!/bin/bash
# copy the measurement set of tools/gpu_round3.sh (gpurun_out/r3) into profiles/round3_*
O=gpurun_out/r3; P=profiles
cp $O/bench.json $P/round3_bench.json
cp $O/prof_bench.json $P/round3_bench_under_rocprof.json
cp $O/prof/r1_kernel_stats.csv $P/round3_kernel_stats.csv
cp $O/pmc_fetch_summary.csv $P/round3_pmc_fetch_size.csv
cp $O/pmc_write_summary.csv $P/round3_pmc_write_size.csv
cp $O/traffic.json $P/round3_traffic.json
for f in layers conv1x1_roofline step_vs_batch conv_lab wino16s_ablation issue_rate mem_rate; do cp $O/$f.txt $P/round3_$f.txt; done
