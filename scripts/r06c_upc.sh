#!/bin/bash
# round 6, second session: the uniformly partitioned form inside the one-trip kernel (the small-call regime's delay-line tail, mid-size calls of 4096 / 8192 frames):
# parity (tests/test_gpu_smallcalls.py, the wire-format cases of both regimes), then the headline chain at 2048- / 4096- / 8192-frame calls with and without it
out=gpurun_out/r06c_upc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_smallcalls.py tests/test_gpu_short.py -x -q > $out/pytest_a.log 2>&1; echo "smallcalls+short rc $? $(tail -1 $out/pytest_a.log)" | tee $out/summary.txt
timeout 900 python -m pytest tests/test_gpu_wire.py -x -q -k "mid_size or small_calls" > $out/pytest_b.log 2>&1; echo "wire rc $? $(tail -1 $out/pytest_b.log)" | tee -a $out/summary.txt
for blk in 2048 4096 8192; do for w in 1 0 1 0; do
  DSP_AMD_CONV_SHORT=$w timeout 300 python bench.py --block $blk --steps 300 --warmup 20 --no-cpu-baseline --no-side-runs > $out/b${blk}_short${w}.json 2> $out/b${blk}_short${w}.err
  python - $out/b${blk}_short${w}.json <<'PY' | tee -a $out/summary.txt
import sys, json
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], round(d['value']), d['unit'], round(d['ms_per_step'], 4), {k: round(v['avg_ms'] * v['launches_per_step'], 4) for k, v in d['roofline']['kernels'].items()}, d['config']['plan'][d['config']['plan'].find('small-calls'):][:120] or d['config']['plan'][d['config']['plan'].find('mid-size'):][:100])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done; done
