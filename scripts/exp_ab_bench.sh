# a / b / a / b of two builds of the library (abso/<name>.so) on one box through bench.py: ARGS = the bench's arguments
for v in "$@" "$@"; do
  cp abso/$v.so dsp_amd/libdsp_amd.so
  python bench.py $ARGS --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
print('$v', round(d['ms_per_step'], 3), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})"
done
