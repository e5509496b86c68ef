#!/bin/bash
# larger transforms at the headline chain: valid fraction 3/4 (N = 2^18), 7/8 (2^19), 15/16 (2^20); rows of 1024 / 2048 / 4096 points
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['config']['plan'][-105:-40])"; }
echo "block 196608 (N=2^18)"; run
echo "block 458752 (N=2^19)"; run --block 458752
echo "block 983040 (N=2^20)"; run --block 983040
echo "block 458752 (N=2^19) pipe off"; DSP_AMD_ROW_PIPE=0 run --block 458752
echo "block 983040 (N=2^20) pipe off"; DSP_AMD_ROW_PIPE=0 run --block 983040
