#!/bin/bash
# Round 4, small measurements: (1) what one product per lock-step could buy (value-only lock-step = the launch chain of a
# stored-K^-1 predictor; the price of forming K^-1 per rebuild); (2) the mahalanobis row kernel at m = 64 -- matrix-pipe
# busy cycles against active cycles.   usage (repo root): bash scripts/r04_extra.sh <outdir under gpurun_out>
set -u
OUT=${1:-gpurun_out/r4extra}
mkdir -p $OUT
for n in 4096 2048; do timeout 300 python scripts/r04_lockstep_bound.py $n 10 10 2>&1 | grep -v amdgpu.ids; done | tee $OUT/lockstep_bound.txt
timeout 120 python scripts/maha_once.py 1250000 64 20 2>&1 | grep -v amdgpu.ids | tee $OUT/maha.txt
timeout 120 python scripts/maha_once.py 1000000 32 20 2>&1 | grep -v amdgpu.ids | tee -a $OUT/maha.txt
bash scripts/r04_pmc.sh $OUT/maha_pmc mahalanobis "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "FETCH_SIZE" -- scripts/maha_once.py 1250000 64 5
