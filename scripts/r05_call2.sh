#!/bin/bash
# round 5, GPU call 2: A/B builds on the headline (non-temporal slab loads in the prepass / first pass; K2 with its filter row from L2 plus a
# rank-20 / rank-10 correction probe), then BASELINE config 4 with the fused first pass and the two-branch two-workgroup K2 switched off in turn
mkdir -p gpurun_out/r05b
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
one() {
  timeout 300 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>gpurun_out/r05b/err.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})
except Exception as e: print('$TAG', 'FAILED', e)"
  tail -2 gpurun_out/r05b/err.log | cut -c1-300
}
{
for v in base f1 f2 f3 base; do cp abso/$v.so dsp_amd/libdsp_amd.so; TAG="headline $v" one; done
cp abso/base.so dsp_amd/libdsp_amd.so
TAG="config4 fused+duo2" one --config 4
TAG="config4 fused, DUO2=0" DSP_AMD_ROW_DUO2=0 one --config 4
TAG="config4 FUSE=0, duo2" DSP_AMD_FUSE=0 one --config 4
TAG="config4 FUSE=0 DUO2=0 (round 4)" DSP_AMD_FUSE=0 DSP_AMD_ROW_DUO2=0 one --config 4
for v in x3 x2; do cp abso/$v.so dsp_amd/libdsp_amd.so; TAG="headline $v (probe: wrong output by design)" one; done
} 2>&1 | tee gpurun_out/r05b/ab.log
