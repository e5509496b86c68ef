#!/bin/bash
# round 5, GPU call 4: the persistent K3 with non-temporal hints on its W loads / slab stores (A/B builds), a / b / a / b
mkdir -p gpurun_out/r05d
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
one() {
  timeout 300 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>gpurun_out/r05d/err.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
for i in 1 2; do for v in base k1 k2 k3; do cp abso/$v.so dsp_amd/libdsp_amd.so; TAG="headline $v" one; done; done
} 2>&1 | tee gpurun_out/r05d/ab.log
