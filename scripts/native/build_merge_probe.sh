#!/bin/bash
# usage (from the repo root; needs elfi_amd/csrc/build/*.o): sh scripts/native/build_merge_probe.sh [extra -D flags]
cd scripts/native
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form "$@" -I../../elfi_amd/csrc -I../../include -c -o /tmp/merge_probe.o merge_probe.hip 2>&1 | grep -E "error" -A3
hipcc --offload-arch=gfx950 -o merge_probe /tmp/merge_probe.o $(ls ../../elfi_amd/csrc/build/*.o | grep -v reject.o)
