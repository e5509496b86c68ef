B10="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}); print('   ', d['config']['plan'])"; }
echo "config 4: biquads + fir_p + resample 96k (block 195584)"; run --block 195584 --chain "$B10 fir_p -t pcm -e double -c 1 {F} resample 96k"
echo "resample only 96k"; run --chain "resample 96k"
echo "resample only 44.1k"; run --chain "resample 44.1k"
echo "config 5-like: 1024 x 2ch hilbert -p 4095 + fir_p 131072"; run --streams 1024 --channels 2 --taps 131072 --block 131072 --chain "hilbert -p 4095 fir_p -t pcm -e double -c 1 {F}"
