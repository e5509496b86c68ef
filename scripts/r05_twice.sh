#!/bin/bash
mkdir -p gpurun_out/r05q
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
for i in 1 2; do
  timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05q/pytest$i.log 2>&1; echo "full suite run $i rc $? : $(tail -1 gpurun_out/r05q/pytest$i.log | cut -c1-200)"
done
