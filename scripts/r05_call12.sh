#!/bin/bash
mkdir -p gpurun_out/r05l
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05l/pytest.log 2>&1; echo "pytest all rc $?"; tail -12 gpurun_out/r05l/pytest.log | cut -c1-300
