# what one rank sees at N = 8, 4, 2, 1 GPUs (256 streams in total): per-rank step time -> strong-scaling estimate
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for s in 32 64 128 256; do echo "streams=$s pipe=auto"; run --streams $s; echo "streams=$s pipe=off"; DSP_AMD_CASCADE_PIPE=0 run --streams $s; done
