# fused against separate kernels at the headline shape for cascades of 1 / 2 / 4 / 6 / 8 sections in front of fir_p(65536)
SEC=("lowpass 1k 0.707" "highshelf 8k 0.7 -3" "eq 100 1.0 3" "eq 200 1.0 -2" "eq 400 2.0 1.5" "eq 800 1.0 -1" "eq 1600 1.4 2" "eq 3200 1.0 -2.5")
for n in ${NSECS:-1 2 4 6 8}; do
  chain=""
  for ((i = 0; i < n; ++i)); do chain="$chain ${SEC[$i]}"; done
  for fuse in 1 0; do
    DSP_AMD_FUSE=$fuse python bench.py --chain "$chain fir_p -t pcm -e double -c 1 {F}" --steps 6 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
print('sections $n fuse $fuse', round(d['ms_per_step'], 3), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})"
  done
done
