#!/bin/bash
mkdir -p gpurun_out/r05i
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05i/pytest.log 2>&1; echo "pytest all rc $?"; tail -30 gpurun_out/r05i/pytest.log | cut -c1-300
