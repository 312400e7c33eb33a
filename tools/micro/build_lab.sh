#!/bin/bash
# Lab build: libp2l_hip_lab.so = the product objects with p2l_wino.hip recompiled under -DP2L_LAB
# (timing ablations + phase trace of the Winograd kernel), and the conv_lab harness against it.
set -e
cd "$(dirname "$0")/../.."
C=pix2latent_amd/csrc
make -C $C -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DP2L_SCALAR_SPLIT -DP2L_LAB -c $C/p2l_wino.hip -o tools/micro/p2l_wino_lab.o
OBJS=$(ls $C/*.o | grep -v p2l_wino.o | grep -v p2l_conv2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/micro/p2l_wino_lab.o -o tools/micro/libp2l_hip_lab.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude -DP2L_LAB tools/micro/conv_lab.cpp -Ltools/micro -lp2l_hip_lab -Wl,-rpath,'$ORIGIN' -o tools/micro/conv_lab_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/micro/conv_lab.cpp -Lpix2latent_amd -lp2l_hip -Wl,-rpath,'$ORIGIN/../../pix2latent_amd' -o tools/micro/conv_lab
