# cascade_rows<1> at 32 streams with parts of the tile traffic switched off (DSP_AMD_CASCADE_DEBUG: 1 = no stores, 2 = no reloads)
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items() if 'casc' in k})"; }
for dbg in 0 1 2 3; do echo "debug=$dbg"; DSP_AMD_CASCADE_DEBUG=$dbg run --streams 32; done
for dbg in 0 3; do echo "G=2 P=8 debug=$dbg"; DSP_AMD_CASCADE_ROWS=208 DSP_AMD_CASCADE_DEBUG=$dbg run --streams 32; done
