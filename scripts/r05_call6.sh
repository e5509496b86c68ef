#!/bin/bash
# round 5, GPU call 6: the whole GPU suite on HEAD; config 4 with the compile-time 17-row instance of the first pass against the run-time-history one; per-rank steps
mkdir -p gpurun_out/r05f
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05f/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05f/pytest.log
one() {
  timeout 300 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
for i in 1 2; do
TAG="config4 HR=17 compile-time" one --config 4
TAG="config4 HR run-time" DSP_AMD_FZ_HR17_RT=1 one --config 4
done
} 2>&1 | tee gpurun_out/r05f/ab.log
bash scripts/exp_scale2.sh 2>&1 | tail -6
