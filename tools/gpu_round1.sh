#!/bin/bash
# one gpurun call: per-kernel parity, model parity, kernel micro-bench
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 2>&1 | tail -60 > gpurun_out/kernels.log
python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 900 2>&1 | tail -60 > gpurun_out/model.log
timeout 300 python tools/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1
echo "=== kernels"; cat gpurun_out/kernels.log | tail -40
echo "=== model"; cat gpurun_out/model.log | tail -40
echo "=== bench"; cat gpurun_out/bench_kernels.log
