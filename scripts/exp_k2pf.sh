# K2 with the next pair's row asked for ahead (DSP_AMD_K2_PF=1; abso/pf8.so, pf12.so: 8 / 12 of the 16 bins) against the resident-filter form, same box
for lib in pf8 pf12 pf8 pf12; do
  cp abso/$lib.so dsp_amd/libdsp_amd.so
  for pf in 1 0; do
    DSP_AMD_K2_PF=$pf python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
print('$lib pf=$pf', round(d['ms_per_step'], 3), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()}, 'parity', d.get('parity', {}).get('rms'))"
  done
done
