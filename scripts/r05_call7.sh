#!/bin/bash
# round 5, GPU call 7: config 5 (1024 x 2 ch, hilbert + zita_convolver 131072 taps): the float32-spectrum instance against fp64 transforms behind the same I/O, and the fp64 fir_p stand-in
mkdir -p gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
one() {
  timeout 300 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()}); print('   ', d['config']['plan'][:400])
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
TAG="config5 f32 spectrum" one --config 5
TAG="config5 ZITA_F64=1" DSP_AMD_ZITA_F64=1 one --config 5
TAG="config5f (fp64 fir_p)" one --config 5f
TAG="config5 f32, ROW_DUO=0" DSP_AMD_ROW_DUO=0 one --config 5
} 2>&1 | tee gpurun_out/r05g/ab.log
