#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in 0 1; do
  P2L_CONV_FORCE=$f python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -k "conv" 2>&1 | tail -8 > gpurun_out/tests_force$f.log
done
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 > gpurun_out/tests_auto.log
for f in -1 0 1; do
  P2L_CONV_FORCE=$f timeout 300 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_kernels_force$f.log
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
for f in 0 1; do echo "=== tests force $f"; cat gpurun_out/tests_force$f.log; done
echo "=== tests auto"; cat gpurun_out/tests_auto.log
paste -d'|' <(cut -c1-36,47-62 gpurun_out/bench_kernels_force-1.log) <(cut -c47-62 gpurun_out/bench_kernels_force0.log) <(cut -c47-62 gpurun_out/bench_kernels_force1.log)
echo "=== bench"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
