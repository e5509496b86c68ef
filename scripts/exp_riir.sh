run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}); print('   ', d['config']['plan'][:300])"; }
echo "linear-phase LR4 crossover low band: forward + reversed Butterworth"; run --chain "lowpass 2k 0.707 lowpass -r 2k 0.707"
echo "reverse highpass 20 (8192 taps)"; run --chain "highpass -r 20 0.707"
