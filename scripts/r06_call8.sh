#!/bin/bash
mkdir -p gpurun_out/c8
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_ladspa.py tests/test_gpu_soak.py tests/test_gpu_dropin.py -m gpu -x -q > gpurun_out/c8/pytest_resident.log 2>&1; echo "resident/ladspa/soak/dropin rc $?: $(tail -1 gpurun_out/c8/pytest_resident.log | cut -c1-200)"; grep -B5 -A25 "^E " gpurun_out/c8/pytest_resident.log | head -80

