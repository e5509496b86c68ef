#!/bin/bash
# the headline chain at the reference's own block size and above: call-size curve (-> profiles/r02_blocksweep.json)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r2sweep}; mkdir -p $O; cd $R
for b in 256 1024 2048 4096 8192 12288 16384 32768 65536 196608; do
  st=$(( 60000000 / b )); [ $st -gt 400 ] && st=400; [ $st -lt 10 ] && st=10
  timeout 300 python bench.py --block $b --steps $st --warmup 16 --no-cpu-baseline > $O/block_$b.json 2> $O/block_$b.err
  python - $O/block_$b.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); r=d['roofline']['kernels']
    print(d['config']['block_frames'], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_ms']*v['launches_per_step'],4) for k,v in r.items()})
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
