#!/bin/bash
# end-of-round record: the default bench line, then rocprofv3 kernel stats of the short bench command
#   gpurun -- 'bash tools/gpu_final_r5.sh'      -> gpurun_out/final/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-extra > $O/bench_under_rocprof.json 2> $O/prof.err
rm -f $O/prof/*kernel_trace.csv
cd $R
python tools/step_vs_batch.py > $O/step_vs_batch.txt 2>&1
echo '# population 18 as the reference chunks 9 + 9: two lanes on two streams, one HIP graph with two branches (the default of the optimizers)' >> $O/step_vs_batch.txt
P2L_TOOL_EXEC=9 P2L_ONLY_N=18 python tools/step_vs_batch.py 2>&1 | tail -1 >> $O/step_vs_batch.txt
echo '# the same chunks one after the other on one stream (P2L_STREAMS=1, eager)' >> $O/step_vs_batch.txt
P2L_STREAMS=1 P2L_TOOL_EXEC=9 P2L_ONLY_N=18 python tools/step_vs_batch.py 2>&1 | tail -1 >> $O/step_vs_batch.txt
head -c 600 $O/bench.json; echo; head -12 $O/prof/*kernel_stats.csv; cat $O/step_vs_batch.txt | tail -9
