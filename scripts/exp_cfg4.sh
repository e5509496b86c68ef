B10="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}); print('   ', d['config']['plan'])"; }
echo "config 4: biquads + fir_p + resample 96k (block 195584)"; run --block 195584 --chain "$B10 fir_p -t pcm -e double -c 1 {F} resample 96k"
echo "config 4 at block 978944"; run --block 978944 --chain "$B10 fir_p -t pcm -e double -c 1 {F} resample 96k"
echo "resample only 96k"; run --block 196608 --chain "resample 96k"
echo "resample only 44.1k"; run --block 196608 --chain "resample 44.1k"
echo "config 5-like: 1024 x 2ch hilbert -p 4095 + fir_p 131072"; run --streams 1024 --channels 2 --taps 131072 --block 131072 --chain "hilbert -p 4095 fir_p -t pcm -e double -c 1 {F}"
echo "config 2: 1 stream x 8 ch, 10 biquads"; run --streams 1 --block 1048576 --chain "$B10"
echo "config 3: 256 x 8 ch fir_p 65536 only (block 196608)"; run --block 196608 --chain "fir_p -t pcm -e double -c 1 {F}"
echo "config 3 at block 983040"; run --chain "fir_p -t pcm -e double -c 1 {F}"
echo "config 5-like at block 917504"; run --streams 1024 --channels 2 --taps 131072 --block 917504 --chain "hilbert -p 4095 fir_p -t pcm -e double -c 1 {F}"
echo "gains among the sections (rows with folded gains)"; run --block 196608 --chain "gain -3 $B10 gain -1 fir_p -t pcm -e double -c 1 {F}"
