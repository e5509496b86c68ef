# cascade_fast variants (CG): 10-biquad chain alone, then the headline chain
B10="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
for cfg in ${CFGS:-0 8 4 2}; do
  echo "DSP_AMD_CASCADE_FAST=$cfg"
  DSP_AMD_CASCADE_FAST=$cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chain "$B10" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('  biquads only', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['digest'])"
  DSP_AMD_CASCADE_FAST=$cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('  headline    ', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()}, d['digest'])"
done
