#!/bin/bash
# power-of-two distances between the buffers of pairs / streams: W (K1 writes, K2, K3 reads 4 pairs at once), the pair rings
# (the cascade writes all of them in step), the streams' input / output slabs
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
echo "all pads off";            DSP_AMD_CONV_WPAD=0 DSP_AMD_CONV_RPAD=0 run --slab-pad 0
echo "W pad only";              DSP_AMD_CONV_RPAD=0 run --slab-pad 0
echo "ring pad only";           DSP_AMD_CONV_WPAD=0 run --slab-pad 0
echo "slab pad only";           DSP_AMD_CONV_WPAD=0 DSP_AMD_CONV_RPAD=0 run
echo "W + ring";                run --slab-pad 0
echo "all (default)";           run
echo "all, slab pad 4";         run --slab-pad 4
echo "all, slab pad 260";       run --slab-pad 260
echo "all, W pad 16";           DSP_AMD_CONV_WPAD=16 run
echo "all, W pad 1040";         DSP_AMD_CONV_WPAD=1040 DSP_AMD_CONV_RPAD=1040 run
