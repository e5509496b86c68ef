B1="lowpass 1k 0.707"
B5="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5"
B10="$B5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
B20="$B10 $B10"
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --chain "$1" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"; }
for dbg in ${DBGS:-0 3}; do
  echo "debug=$dbg (1 = no stores, 2 = no reloads)"; export DSP_AMD_CASCADE_DEBUG=$dbg
  for c in "gain -3" "$B1" "$B5" "$B10" "$B20"; do run "$c"; done
done
