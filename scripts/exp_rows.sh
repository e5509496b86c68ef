# cascade_rows<G> with P waves per group (DSP_AMD_CASCADE_ROWS = 100 G + P; 0 = off -> cascade_wave) at per-rank stream counts
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items() if 'casc' in k})"; }
for s in 32 64; do for e in 0 204 206 208; do echo "streams=$s ROWS=$e"; DSP_AMD_CASCADE_ROWS=$e run --streams $s; done; done
for e in 408 406 208 204; do echo "streams=128 ROWS=$e"; DSP_AMD_CASCADE_ROWS=$e run --streams 128; done
for e in 404 405 204; do echo "streams=256 ROWS=$e"; DSP_AMD_CASCADE_ROWS=$e run --streams 256; done
