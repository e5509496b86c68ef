#!/bin/bash
# round 5, final GPU call: the whole GPU suite on HEAD, then the round's profile set (scripts/profile_round.sh) and the per-rank steps
mkdir -p gpurun_out/r05final
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
AMD_LOG_LEVEL=1 timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05final/pytest.log 2>&1; echo "pytest all rc $?"; tail -3 gpurun_out/r05final/pytest.log | cut -c1-300
bash scripts/profile_round.sh r05c > gpurun_out/r05final/profile.log 2>&1; tail -3 gpurun_out/r05final/profile.log
bash scripts/exp_scale2.sh 2>&1 | tail -6
