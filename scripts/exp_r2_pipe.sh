#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O; cd $R
run() { timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for w in 256 512 128 1024; do echo "pipe wgs=$w"; DSP_AMD_ROW_PIPE_WGS=$w run; done
echo "pipe off"; DSP_AMD_ROW_PIPE=0 run
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/b1.json 2> $O/stats.err
DSP_AMD_ROW_PIPE=0 rocprofv3 --kernel-trace --stats -d $O/stats0 -o stats --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/b0.json 2>> $O/stats.err
head -8 $O/stats/stats_kernel_stats.csv | cut -c1-200; head -8 $O/stats0/stats_kernel_stats.csv | cut -c1-200
find $O -name "*trace.csv" -size +5M -delete
