run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for dbg in 0 1 2 3 4 7; do echo "wave debug=$dbg (1 = no stores, 2 = no loads, 4 = no compute)"; DSP_AMD_CASCADE_DEBUG=$dbg run; done
