#!/bin/bash
# round 6, last call: the round's profile set on HEAD (r06b), then N consecutive one-process runs of the GPU suite on HEAD
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
bash scripts/profile_round.sh r06b > gpurun_out/final/profile_round.log 2>&1; tail -8 gpurun_out/final/profile_round.log | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/r06b/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})
for k,v in d.get('side_runs',{}).get('other_configs',{}).items(): print(k, round(v.get('value',0)), round(v.get('ms_per_step',0),2))
"
rm -f gpurun_out/verify/summary.txt
scripts/r06_verify.sh ${1:-8} > gpurun_out/final/verify.txt 2>&1; cat gpurun_out/final/verify.txt | cut -c1-160
