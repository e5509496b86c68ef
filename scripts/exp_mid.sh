#!/bin/bash
# small and mid-size calls (round 3): the tests, then the headline chain at 256 ... 4096-frame calls, default plan against the switches in $EXTRA
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_smallcalls.py tests/test_gpu_wire.py tests/test_gpu_fallbacks.py -x -q -k "mid_size or small_calls or UPC or FDL" 2>&1 | tail -8
run() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']
print(round(d['value']/1e3,2), round(d['ms_per_step'],4), {k:round(v['avg_ms']*v['launches_per_step'],4) for k,v in r.items()})"; }
for b in ${BLOCKS:-256 1024 2048 4096}; do
for e in "" $EXTRA; do echo "block $b $e"; env $e timeout 300 python bench.py --block $b --steps 300 --warmup 16 --no-cpu-baseline 2>/dev/null | run; done; done
