#!/bin/bash
# A/B build of the WHOLE library with extra -D flags: tools/ab_build_all.sh -DP2L_SUB_RESID
# -> tools/micro/libp2l_hip_ab.so (load it through P2L_LIB_PATH, see pix2latent_amd/_native.py)
set -e
cd "$(dirname "$0")/.."
C=pix2latent_amd/csrc
O=tools/micro/ab_obj
mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result -Wno-unused-command-line-argument"
for f in $C/*.hip; do
  n=$(basename $f .hip); X=""
  if [ "$n" = p2l_wino ]; then X="-Xclang -target-feature -Xclang -packed-fp32-ops -DP2L_SCALAR_SPLIT"; fi
  ( /opt/rocm/bin/hipcc $FL $X "$@" -c $f -o $O/$n.o 2> >(grep -v "not a recognized feature" >&2) ) &
done
/opt/rocm/bin/hipcc $FL "$@" -x hip -c $C/p2l_api.cpp -o $O/p2l_api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o tools/micro/libp2l_hip_ab.so
echo built tools/micro/libp2l_hip_ab.so
