#!/bin/bash
# the driver's own command (tests/conftest.py gives every GPU module a process of its own)
mkdir -p gpurun_out/r05iso
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
t=$(date +%s)
timeout 400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05iso/pytest.log 2>&1; echo "rc $? ($(( $(date +%s) - t )) s): $(tail -1 gpurun_out/r05iso/pytest.log | cut -c1-200)"
grep -n "^FAILED\|^ERROR\|died in this test" gpurun_out/r05iso/pytest.log | head -5 | cut -c1-250
