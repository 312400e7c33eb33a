#!/bin/bash
# full GPU gate: every -m gpu test, the default bench line (with cpu_baseline), then
# rocprofv3 kernel stats + the two PMC passes over the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > gpurun_out/tests.log
echo "=== tests"; cat gpurun_out/tests.log
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "=== bench"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
bash tools/gpu_profile.sh > gpurun_out/profile.log 2>&1
tail -40 gpurun_out/profile.log
