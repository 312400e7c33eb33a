#!/bin/bash
# GPU gate of a round: every -m gpu test (verbose failures), then the default bench line.
# usage (from the container): gpurun --timeout 1500 -- 'bash tools/gpu_gate.sh [pytest args]'
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q --timeout 900 -rf --durations=15 "$@" ) > gpurun_out/tests.log 2>&1
echo "=== tests"; tail -60 gpurun_out/tests.log
( time python bench.py ) > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "=== bench"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
