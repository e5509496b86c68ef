#!/bin/bash
# cascade_rows<4>: waves per group (P) at the headline shape
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for p in 404 405 406 407 408 204 208; do echo "ROWS=$p"; DSP_AMD_CASCADE_ROWS=$p run; done
