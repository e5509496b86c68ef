#!/bin/bash
mkdir -p gpurun_out/c6
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash scripts/r06_resident_timing.sh > gpurun_out/c6/resident_timing.txt 2>&1; cat gpurun_out/c6/resident_timing.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_short.py tests/test_gpu_wire.py tests/test_gpu_conv.py -m gpu -x -q > gpurun_out/c6/pytest_short.log 2>&1; echo "short/wire/conv rc $?: $(tail -1 gpurun_out/c6/pytest_short.log | cut -c1-200)"
bash scripts/r06_fz_counters.sh r06_fdl_2048 --block 2048 > gpurun_out/c6/fdl.txt 2>&1; tail -3 gpurun_out/c6/fdl.txt | cut -c1-200
