#!/bin/bash
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "register_resident or direct_kernel_fp16x2 or promised_maxima or fused_affine_relu_bwd" > $O/h2r_tests.txt 2>&1
tail -15 $O/h2r_tests.txt
timeout 600 python tools/bench_h2r.py > $O/bench_h2r.txt 2> $O/bench_h2r.err
cat $O/bench_h2r.txt; tail -3 $O/bench_h2r.err
timeout 900 python -m pytest tests/test_sg2_fullsize_oracle_gpu.py tests/test_lanes_gpu.py -q -k "out_of_memory" -s > $O/fix_tests.txt 2>&1
grep "cars-512\|passed\|failed" $O/fix_tests.txt
