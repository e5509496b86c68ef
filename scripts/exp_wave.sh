# cascade_wave vs cascade_fast at the per-rank stream counts of 8, 4, 2, 1 GPUs
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for s in 32 64 128; do echo "streams=$s wave=auto"; run --streams $s; echo "streams=$s wave=off"; DSP_AMD_CASCADE_WAVE=0 run --streams $s; done
echo "streams=128 wave=402"; DSP_AMD_CASCADE_WAVE=402 run --streams 128
echo "streams=128 wave=205"; DSP_AMD_CASCADE_WAVE=205 run --streams 128
echo "streams=64 wave=110"; DSP_AMD_CASCADE_WAVE=110 run --streams 64
echo "streams=64 wave=203"; DSP_AMD_CASCADE_WAVE=203 run --streams 64
echo "streams=32 wave=105"; DSP_AMD_CASCADE_WAVE=105 run --streams 32
echo "streams=256 fast"; run --streams 256
echo "streams=256 wave=801"; DSP_AMD_CASCADE_WAVE=801 run --streams 256
echo "streams=256 wave=401"; DSP_AMD_CASCADE_WAVE=401 run --streams 256
echo "streams=256 wave=402"; DSP_AMD_CASCADE_WAVE=402 run --streams 256
