#!/bin/bash
# round 6, second session: the one-trip convolver with two exchanges (radix 32 / 16 / 16) and its 16384-point window (radix 32 / 32 / 16):
# parity first (tests/test_gpu_short.py, scripts/soak_short.py), then BASELINE config 5 with either window forced, a / b / a / b on one box.
out=gpurun_out/r06c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python -m pytest tests/test_gpu_short.py -x -q > $out/pytest_short.log 2>&1; echo "pytest_short rc $?" | tee $out/summary.txt
timeout 600 python scripts/soak_short.py ${SOAK:-24} > $out/soak_short.log 2>&1; echo "soak rc $? $(tail -1 $out/soak_short.log)" | tee -a $out/summary.txt
for i in 1 2; do for w in 13 14; do
  DSP_AMD_CONV_SHORT=$w timeout 300 python bench.py --config 5 --steps 5 --no-cpu-baseline > $out/c5_w${w}_$i.json 2> $out/c5_w${w}_$i.err
  python - $out/c5_w${w}_$i.json <<'PY' | tee -a $out/summary.txt
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], round(d['value']), d['unit'], {k: round(v['avg_ms'] * v['launches_per_step'], 2) for k, v in d['roofline']['kernels'].items()})
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done; done
