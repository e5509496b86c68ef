run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
F="fir_p -t pcm -e double -c 1 {F}"
B1="lowpass 1k 0.707"; B3="$B1 eq 100 1.0 3 eq 200 1.0 -2"; B5="$B3 eq 400 2.0 1.5 eq 800 1.0 -1"; B7="$B5 eq 1600 1.4 2 eq 3200 1.0 -2.5"
for c in "$B1" "$B3" "$B5" "$B7"; do for cg in 8 4; do echo "n=$(echo $c | wc -w) CG=$cg"; DSP_AMD_CASCADE_FAST=$cg run --chain "$c $F"; done; done
