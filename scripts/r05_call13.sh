#!/bin/bash
mkdir -p gpurun_out/r05m
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1200 python scripts/soak_short.py ${SEEDS:-60} > gpurun_out/r05m/soak_short.log 2>&1; echo "soak_short rc $?"; tail -5 gpurun_out/r05m/soak_short.log | cut -c1-300; echo "one-trip plans:" $(grep -c "one-trip True" gpurun_out/r05m/soak_short.log) "of" $(grep -c "^seed" gpurun_out/r05m/soak_short.log)
