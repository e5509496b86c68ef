#!/bin/bash
# round 5, GPU call 10: conv_short with the next window prefetched (filter row from L2) against the plain form; its tests
mkdir -p gpurun_out/r05j
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_short.py tests/test_gpu_wire.py -m gpu -x -q > gpurun_out/r05j/pytest_short.log 2>&1; echo "pytest short+wire rc $?"; tail -4 gpurun_out/r05j/pytest_short.log
DSP_AMD_SHORT_PF=0 timeout 900 python -m pytest tests/test_gpu_short.py -m gpu -x -q > gpurun_out/r05j/pytest_short_nopf.log 2>&1; echo "pytest short (PF=0) rc $?"; tail -2 gpurun_out/r05j/pytest_short_nopf.log
one() {
  timeout 300 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
for i in 1 2; do
TAG="config5 prefetch" one --config 5
TAG="config5 PF=0" DSP_AMD_SHORT_PF=0 one --config 5
done
} 2>&1 | tee gpurun_out/r05j/ab.log
