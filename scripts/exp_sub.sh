run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['digest']['sum_of_squares'])"; }
echo base; DSP_AMD_ROW_G=1 run
echo "row G=4"; DSP_AMD_ROW_G=4 run
export DSP_AMD_ROW_G=1
for mb in 16 32 64; do for ns in 2 4 8; do echo "chunk $mb MB x $ns substreams"; DSP_AMD_CONV_CHUNK_MB=$mb DSP_AMD_CONV_SUBSTREAMS=$ns run; done; done
