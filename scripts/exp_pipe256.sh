run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for c in 0 402 403 205 203 110; do echo "S=256 DSP_AMD_CASCADE_PIPE=$c"; DSP_AMD_CASCADE_PIPE=$c run; done
echo "config 2: 1 stream x 8 ch, 10 biquads"; B10="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
run --streams 1 --chain "$B10"; DSP_AMD_CASCADE_PIPE=0 run --streams 1 --chain "$B10"
