#!/bin/bash
mkdir -p gpurun_out
for a in 0 1 2 3; do
  echo "=== P2L_ABL=$a (1: no B loads, 2: no transform)"
  P2L_ABL=$a python tools/bench_bf3.py 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3, $(NF-8),$(NF-7),$(NF-6),$(NF-5)}'
done | tee gpurun_out/abl_wino.txt
