#!/bin/bash
# round 5, GPU call 8: the one-trip convolver -- its own tests, then the whole suite (short filters everywhere now take it), config 5 before / after
mkdir -p gpurun_out/r05h
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_short.py -m gpu -x -q > gpurun_out/r05h/pytest_short.log 2>&1; echo "pytest short rc $?"; tail -15 gpurun_out/r05h/pytest_short.log
one() {
  timeout 300 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()}); print('   ', d['config']['plan'][:300])
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
TAG="config5 one-trip hilbert" one --config 5
TAG="config5 CONV_SHORT=0" DSP_AMD_CONV_SHORT=0 one --config 5
} 2>&1 | tee gpurun_out/r05h/ab.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05h/pytest.log 2>&1; echo "pytest all rc $?"; tail -30 gpurun_out/r05h/pytest.log | cut -c1-300
