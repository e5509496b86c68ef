run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for s in 256 128; do
for d in 0 1 2 3; do echo "streams=$s rows debug=$d (1 = no stores, 2 = no loads)"; DSP_AMD_CASCADE_DEBUG=$d run --streams $s; done
done
for P in 2 3 4 5 6 8; do echo "streams=256 P=$P"; DSP_AMD_CASCADE_ROWS=$P run --streams 256; done
for P in 4 6 8; do echo "streams=128 P=$P"; DSP_AMD_CASCADE_ROWS=$P run --streams 128; done
for P in 4 8; do echo "streams=64 P=$P"; DSP_AMD_CASCADE_ROWS=$P run --streams 64; done
