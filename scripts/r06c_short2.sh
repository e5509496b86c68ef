#!/bin/bash
# round 6, second session, second call: the one-trip convolver up to 8193 taps -- parity (tests, soak), then an 8000-tap filter on config 5's shape
# (1024 x 2 ch, 917504-frame calls) one-trip against the four-step transforms, then the counters of config 5's step with the new conv_short.
out=gpurun_out/r06c2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python -m pytest tests/test_gpu_short.py -x -q > $out/pytest_short.log 2>&1; echo "pytest_short rc $? $(tail -1 $out/pytest_short.log)" | tee $out/summary.txt
timeout 900 python scripts/soak_short.py ${SOAK:-40} > $out/soak_short.log 2>&1; echo "soak rc $? $(tail -1 $out/soak_short.log)" | tee -a $out/summary.txt
for i in 1 2; do for w in 1 0; do
  DSP_AMD_CONV_SHORT=$w timeout 300 python bench.py --streams 1024 --channels 2 --block 917504 --taps 8000 --chain "fir_p -t pcm -e double -c 1 {F}" --steps 5 --no-cpu-baseline > $out/t8000_short${w}_$i.json 2> $out/t8000_short${w}_$i.err
  python - $out/t8000_short${w}_$i.json <<'PY' | tee -a $out/summary.txt
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], round(d['value']), d['unit'], {k: round(v['avg_ms'] * v['launches_per_step'], 2) for k, v in d['roofline']['kernels'].items()}, d['config']['plan'][:160])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done; done
bash scripts/r06_fz_counters.sh r06c2/ctr --config 5 > $out/counters_table.txt 2>&1
tail -3 $out/counters_table.txt
