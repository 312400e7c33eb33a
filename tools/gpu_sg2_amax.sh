#!/bin/bash
# StyleGAN2 configurations with and without the maxima hand-over of the plan (P2L_AMAX=0: every launch its own pass)
mkdir -p gpurun_out/sg2amax
O=gpurun_out/sg2amax
for cfg in c5 c4; do
  for am in 1 0; do
    P2L_AMAX=$am timeout 600 python tools/step_sg2_one.py $cfg $O/layers_${cfg}_amax$am.txt 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/P2L_AMAX=$am /"
  done
done | tee $O/summary.txt
timeout 900 python -m pytest tests/test_sg2_fullsize_oracle_gpu.py -q 2>&1 | tail -2 | tee -a $O/summary.txt
