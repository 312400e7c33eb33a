#!/bin/bash
# A/B build: a second libp2l_hip.so with extra -D flags for ONE translation unit, loaded through
# P2L_LIB_PATH (read by pix2latent_amd/_native.py):   tools/ab_build.sh p2l_wino -DP2L_NO_WINO_SLICES
set -e
cd "$(dirname "$0")/.."
C=pix2latent_amd/csrc
TU=$1; shift
make -C $C -j8 >/dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result"
if [ "$TU" = p2l_wino ]; then FL="$FL -Xclang -target-feature -Xclang -packed-fp32-ops -DP2L_SCALAR_SPLIT"; fi
mkdir -p tools/micro
/opt/rocm/bin/hipcc $FL "$@" -c $C/$TU.hip -o tools/micro/${TU}_ab.o 2> >(grep -v "not a recognized feature" >&2)
OBJS=$(ls $C/*.o | grep -v "$C/$TU.o" | grep -v p2l_conv2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/micro/${TU}_ab.o -o tools/micro/libp2l_hip_ab.so
echo built tools/micro/libp2l_hip_ab.so
