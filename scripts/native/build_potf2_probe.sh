#!/bin/bash
# usage: build_potf2_probe.sh <out name> [extra -D flags, e.g. -DELFIHIP_POTF2_STAMP for the cycle stamps]
# (run from the repo root; needs elfi_amd/csrc/build/*.o)
out=$1; shift
cd scripts/native
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form "$@" -I../../elfi_amd/csrc -I../../include -c -o /tmp/$out.o potf2_probe.hip 2>&1 | grep -E "error" 
hipcc --offload-arch=gfx950 -o $out /tmp/$out.o $(ls ../../elfi_amd/csrc/build/*.o | grep -v gp_fit.o)
