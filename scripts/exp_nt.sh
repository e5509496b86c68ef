run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for m in 0 1 2 4 8 16 32 63; do echo "DSP_AMD_CONV_NT=$m"; DSP_AMD_CONV_NT=$m run; done
