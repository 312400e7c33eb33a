#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25 > gpurun_out/tests.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "=== tests"; cat gpurun_out/tests.log
echo "=== bench"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
