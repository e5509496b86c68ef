# A/B of an environment switch on the same box: $1 = VAR, then the values in turn (a b a b); DBG_MODES as in exp_ab_so.sh
var=$1; shift
for v in "$@" "$@"; do
  echo "== $var=$v"
  env $var=$v python scripts/exp_fused.py dbg $DBG_MODES 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if '{' in l:
        n, j = l.split(' ', 1); d = json.loads(j); print(n, d['ms_per_step'], 'fused_col_fwd', d['kernels_ms'].get('fused_col_fwd'))
    else: print(l, end='')
"
done
