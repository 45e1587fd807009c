#!/bin/bash
# usage (from the repo root; needs elfi_amd/csrc/build/*.o): sh scripts/native/build_sel_probe.sh
cd scripts/native
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DELFIHIP_SEL_STAMP -I../../elfi_amd/csrc -I../../include -c -o /tmp/sel_probe.o sel_probe.hip 2>&1 | grep -E "error" -A3
hipcc --offload-arch=gfx950 -o sel_probe /tmp/sel_probe.o $(ls ../../elfi_amd/csrc/build/*.o | grep -v topk.o)
