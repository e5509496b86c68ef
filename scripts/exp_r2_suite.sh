#!/bin/bash
# GPU test suite + default bench (+ optional extra command)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r2s}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],3), d['inner_repeats'], [round(x,3) for x in d['ms_per_step_of_each_region']], 'drain', round(d['ms_drain'],2), d['drain_frames'])
print({k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r['kernels'].items()}, 'copy', round(r['measured_copy_GBps']))
print(json.dumps(d['cpu_baseline'], indent=1)[:3000])
PY
