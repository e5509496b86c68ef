#!/bin/bash
# conv tests + short default bench; extra args are passed to bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r2q}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6
run() { timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['config']['plan'][-120:])"; }
echo "default"; run "$@"
echo "pipe off"; DSP_AMD_ROW_PIPE=0 run "$@"
