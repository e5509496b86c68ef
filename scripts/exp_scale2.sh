# per-rank kernel split at the stream counts one rank sees at N = 8, 4, 2, 1 GPUs
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['config']['plan'])"; }
for s in 32 64 128 256; do echo "streams=$s"; run --streams $s; done
