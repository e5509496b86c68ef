#!/bin/bash
# round 2, first GPU call: HBM ceiling of the access patterns, the hipFFT/rocFFT library route on the headline problem,
# and the block-size curve of the headline chain as it stood at the start of the round
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; mkdir -p $O; cd $R
timeout 300 scripts/ubench/hbmprobe 4 > $O/hbmprobe.txt 2>&1
timeout 400 scripts/ubench/rocfft_baseline 18 1024 > $O/rocfft_baseline.json 2> $O/rocfft.err
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/rocfft_stats -o stats --output-format csv -- $R/scripts/ubench/rocfft_baseline 18 1024 > $O/rocfft_under_rocprof.json 2>> $O/rocfft.err )
for b in 2048 16384 65536 196608; do
  st=$(( 40000000 / b )); [ $st -gt 200 ] && st=200; [ $st -lt 10 ] && st=10
  timeout 300 python bench.py --block $b --steps $st --warmup 3 --no-cpu-baseline > $O/block_$b.json 2> $O/block_$b.err
done
find $O -name "*.csv" -size +5M -delete
cat $O/hbmprobe.txt; cat $O/rocfft_baseline.json; for b in 2048 16384 65536 196608; do python - $O/block_$b.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); r=d['roofline']['kernels']
    print(d['config']['block_frames'], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_ms']*v['launches_per_step'],4) for k,v in r.items()})
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
