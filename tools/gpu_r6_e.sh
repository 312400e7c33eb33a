#!/bin/bash
# maxima compaction: SG2 configs before/after numbers come from comparing with profiles/round6_sg2_*_layers.txt
mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e
cd $R
python tools/step_sg2_one.py c5 $O/sg2_c5_layers.txt 2>/dev/null | head -1
python tools/step_sg2_one.py c4 $O/sg2_c4_layers.txt 2>/dev/null | head -1
head -14 $O/sg2_c5_layers.txt | tail -11
timeout 1500 python -m pytest tests/test_sg2_fullsize_oracle_gpu.py tests/test_fullsize_gpu.py tests/test_shard_bits_gpu.py tests/test_rank_parity_gpu.py tests/test_pipeline_gpu.py -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
