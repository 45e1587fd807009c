#!/bin/bash
# usage (from the repo root; needs elfi_amd/csrc/build/*.o): sh scripts/native/build_ada_probe.sh
cd scripts/native
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DELFIHIP_ADA_STAMP -I../../elfi_amd/csrc -I../../include -c -o /tmp/ada_probe.o ada_probe.hip 2>&1 | grep -E "error" -A3
hipcc --offload-arch=gfx950 -o ada_probe /tmp/ada_probe.o $(ls ../../elfi_amd/csrc/build/*.o | grep -v adaptive.o)
