# A/B of two builds of the library (copied over dsp_amd/libdsp_amd.so in turn): cascade time per launch at the per-rank
# stream counts; the last one named stays in place
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items() if 'casc' in k})"; }
for s in 32 64 128 256; do for lib in "$@"; do echo "streams=$s lib=$lib"; cp $lib dsp_amd/libdsp_amd.so; run --streams $s; done; done
