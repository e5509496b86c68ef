#!/bin/bash
# round 5: the soak of the fused first pass over random shapes (history rows 1 .. 32, with and without `resample 96k`, with and without sections)
mkdir -p gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1200 python scripts/soak_fused.py ${SEEDS:-60} > gpurun_out/r05e/soak.log 2>&1; echo "soak rc $?"; tail -4 gpurun_out/r05e/soak.log; echo "fused plans:" $(grep -c "fused-plan True" gpurun_out/r05e/soak.log) "of" $(grep -c "^seed" gpurun_out/r05e/soak.log)
