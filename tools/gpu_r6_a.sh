#!/bin/bash
# round 6, call A: the new tests of this round + this round's baseline numbers on one box
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a
cd $R
timeout 1500 python -m pytest tests/test_sg2_fullsize_oracle_gpu.py -x -q -s -m gpu > $O/sg2_fullsize.txt 2>&1
echo "sg2 fullsize rc=$?" >> $O/sg2_fullsize.txt
timeout 600 python -m pytest tests/test_lanes_gpu.py tests/test_abi.py tests/test_loss_cache_gpu.py tests/test_sublanes_gpu.py -x -q > $O/small_tests.txt 2>&1
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "another_format or subpixel" >> $O/small_tests.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
python tools/prof_layers.py > $O/layers.txt 2>/dev/null
python tools/step_vs_batch.py 2>/dev/null | grep "local candidates" > $O/step_vs_batch.txt
tail -5 $O/sg2_fullsize.txt; tail -3 $O/small_tests.txt; cut -c1-400 $O/bench_short.json; tail -2 $O/layers.txt; cat $O/step_vs_batch.txt
