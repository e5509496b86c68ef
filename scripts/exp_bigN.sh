run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['config']['plan'][100:190])"; }
echo "block 196608 (N=2^18)"; run
echo "block 458752 (N=2^19)"; run --block 458752
echo "block 983040 (N=2^20)"; run --block 983040
for mb in 64 128 192 256; do echo "block 983040 chunk $mb MB"; DSP_AMD_CONV_CHUNK_MB=$mb run --block 983040; done
for mb in 64 128 192; do echo "block 196608 chunk $mb MB"; DSP_AMD_CONV_CHUNK_MB=$mb run; done
