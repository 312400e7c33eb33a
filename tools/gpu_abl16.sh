#!/bin/bash
# timing ablations of the 16x16-pixel Winograd kernel ($P2L_ABL bits: 1 no weight loads, 2 no
# transform, 4 no patch staging, 8 no MFMAs, 16 no split; results are wrong when set)
for a in 0 1 2 4 8 16 3 19 27; do
  echo "=== P2L_ABL=$a"
  P2L_ABL=$a python tools/bench_bf3.py 2>&1 | grep -v amdgpu | sed -E 's/f32 .*wino16/wino16/' | cut -c1-80
done
