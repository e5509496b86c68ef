#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the round's changes; K2 A/B builds (filter row from L2; + a rank-20 / rank-10 correction probe);
# BASELINE config 4 with the fused first pass and the two-branch two-workgroup K2, each switched off in turn
mkdir -p gpurun_out/r05a
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r05a/pytest.log
tail -5 gpurun_out/r05a/pytest.log
one() {  # name, env..., -- bench args
  python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
    print('$TAG', round(d['ms_per_step'], 3), round(d['value'] / 1e3, 2), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})
except Exception as e: print('$TAG', 'FAILED', e)"
}
{
for v in base x1 x2 x3 base; do cp abso/$v.so dsp_amd/libdsp_amd.so; TAG="headline $v" one; done
cp abso/base.so dsp_amd/libdsp_amd.so
TAG="config4 fused+duo2" one --config 4
TAG="config4 fused, DUO2=0" DSP_AMD_ROW_DUO2=0 one --config 4
TAG="config4 FUSE=0, duo2" DSP_AMD_FUSE=0 one --config 4
TAG="config4 FUSE=0 DUO2=0 (round 4)" DSP_AMD_FUSE=0 DSP_AMD_ROW_DUO2=0 one --config 4
} 2>&1 | tee gpurun_out/r05a/ab.log
