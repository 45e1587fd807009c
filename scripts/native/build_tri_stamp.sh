#!/bin/sh
# Developer probe: libelfihip with wall-clock stamps in the triangular products (ELFIHIP_TRI_STAMP), all symbols exported,
# built beside the product library as scripts/native/libelfihip_stamp.so (git-ignored).  Used by scripts/tri_timeline.py.
set -e
cd "$(dirname "$0")/../../elfi_amd/csrc"
mkdir -p build_stamp
for f in *.hip; do
  o=build_stamp/${f%.hip}.o
  if [ "$f" = gp_predict.hip ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DELFIHIP_TRI_STAMP -c $f -o $o
  else
    cp build/${f%.hip}.o $o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/native/libelfihip_stamp.so build_stamp/*.o
