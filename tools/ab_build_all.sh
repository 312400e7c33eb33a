#!/bin/bash
# A/B build of the WHOLE library with extra -D flags (for switches in shared headers):
#   tools/ab_build_all.sh out.so -DP2L_NT_STORE     -> tools/micro/out.so  (P2L_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
C=pix2latent_amd/csrc
OUT=$1; shift
D=tools/micro/ab_all; mkdir -p $D
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-result -Wno-unused-command-line-argument"
for f in $C/*.hip; do
  b=$(basename $f .hip); X=""
  [ "$b" = p2l_conv2 ] && continue
  [ "$b" = p2l_wino ] && X="-Xclang -target-feature -Xclang -packed-fp32-ops -DP2L_SCALAR_SPLIT"
  /opt/rocm/bin/hipcc $FL $X "$@" -c $f -o $D/$b.o 2> >(grep -v "not a recognized feature" >&2) &
done
/opt/rocm/bin/hipcc $FL "$@" -x hip -c $C/p2l_api.cpp -o $D/p2l_api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o tools/micro/$OUT
echo built tools/micro/$OUT
