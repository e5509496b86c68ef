#!/bin/bash
# round 5, GPU call 3: K2 with 8 points per thread (DSP_AMD_ROW_OCT=1): parity at the headline's size, then a / b / a / b on the headline and config 3
mkdir -p gpurun_out/r05c
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
DSP_AMD_ROW_OCT=1 timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "bench_default or config3_hop" > gpurun_out/r05c/pytest_oct.log 2>&1; echo "pytest (ROW_OCT=1) rc $?"; tail -3 gpurun_out/r05c/pytest_oct.log
one() {
  timeout 300 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>gpurun_out/r05c/err.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
for i in 1 2; do
TAG="headline duo" one
TAG="headline oct" DSP_AMD_ROW_OCT=1 one
done
TAG="config3 duo" one --config 3
TAG="config3 oct" DSP_AMD_ROW_OCT=1 one --config 3
} 2>&1 | tee gpurun_out/r05c/ab.log
