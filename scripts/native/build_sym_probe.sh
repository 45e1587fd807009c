#!/bin/bash
# usage (from the repo root): sh scripts/native/build_sym_probe.sh
cd scripts/native
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o sym_probe sym_probe.hip 2>&1 | grep -E "error|warning: v" -A3
