#!/bin/bash
# cascade_rows<1> (256 channels on the GPU) at different channel counts per stream: the stride of a wave's 8-byte tile accesses
# is the frame size (64 / 32 / 16 / 8 bytes) -- what do the strided accesses cost?
cd $GRAFT_REPO_ROOT
for sc in "32 8" "64 4" "128 2" "256 1"; do set -- $sc
 python bench.py --streams $1 --channels $2 --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']
print('$1 x $2', round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['config']['plan'][:90])"
done
