# A/B of two builds of the library on the same box: abso/<name>.so copied over dsp_amd/libdsp_amd.so in turn; DBG_MODES = debug instances to time as well
for v in "$@"; do
  cp abso/$v.so dsp_amd/libdsp_amd.so
  echo "== $v"
  python scripts/exp_fused.py dbg $DBG_MODES 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if '{' in l:
        n, j = l.split(' ', 1); d = json.loads(j); print(n, d['ms_per_step'], 'fused_col_fwd', d['kernels_ms'].get('fused_col_fwd'))
    else: print(l, end='')
"
done
