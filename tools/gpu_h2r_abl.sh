#!/bin/bash
# ablation builds of the register-resident kernel, timed on one box
mkdir -p gpurun_out/h2r_abl
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/h2r_abl
cd $R
python tools/bench_h2r_abl.py 2>/dev/null | tee $O/abl.txt
for a in 1 2 8 16 32 33 49 57 59; do
  P2L_LIB_PATH=$R/tools/micro/libp2l_hip_abl$a.so python tools/bench_h2r_abl.py 2>/dev/null | sed "s/^/ABL=$a /" | tee -a $O/abl.txt
done
