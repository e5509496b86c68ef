#!/bin/bash
# round 6, second session: the whole GPU suite as one process on HEAD, then the default bench line
out=gpurun_out/r06c_suite; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
now=$(date +%s)
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1
echo "pytest rc $? $(( $(date +%s) - now )) s: $(tail -1 $out/pytest.log)" | tee $out/summary.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import sys, json
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('headline', round(d['value']), d['unit'], d['ms_per_step'], 'frac', round(d['roofline']['frac'], 3))
for k, v in d['side_runs']['other_configs'].items(): print(' ', k, round(v.get('value', 0)), {a: round(b['ms_per_step'], 2) for a, b in v.get('kernels', {}).items()})
PY
