#!/bin/bash
# round 6, call 5: the resident wave with the settled word; LADSPA block times; the one-trip convolver with two point sets per thread (parity, then A/B on
# config 5); the suite in one process without the slack behind device buffers, and once with every buffer poisoned (DSP_AMD_GUARD=5)
mkdir -p gpurun_out/c5
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_ladspa.py tests/test_gpu_soak.py -m gpu -x -q > gpurun_out/c5/pytest_resident.log 2>&1; echo "resident/ladspa/soak rc $?: $(tail -1 gpurun_out/c5/pytest_resident.log | cut -c1-200)"
{ echo "== default (request mailbox in device memory)"; bash scripts/exp_ladspa_rate.sh 2>&1 | grep -v "^$"; echo "== DSP_AMD_PLUGIN_MAILBOX=host"; DSP_AMD_PLUGIN_MAILBOX=host bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so DSP_AMD_PLUGIN_MAPPED_KB=32"; echo "== DSP_AMD_PLUGIN_RESIDENT=0"; DSP_AMD_PLUGIN_RESIDENT=0 bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so DSP_AMD_PLUGIN_MAPPED_KB=32"; echo "== crossover config"; bash scripts/exp_ladspa_rate_xover.sh 2>&1; } > gpurun_out/c5/ladspa_rate.txt 2>&1
grep -A1 "^==\|gpu.so" gpurun_out/c5/ladspa_rate.txt | grep -o "^==.*\|ladspa_dsp_[a-z]*.so.*\|run_seconds.*" | cut -c1-200
DSP_AMD_SHORT_VT=2 timeout 600 python -m pytest tests/test_gpu_short.py tests/test_gpu_wire.py -m gpu -x -q > gpurun_out/c5/pytest_short_vt2.log 2>&1; echo "short VT=2 rc $?: $(tail -1 gpurun_out/c5/pytest_short_vt2.log | cut -c1-200)"
for rep in 1 2; do for vt in 1 2; do
  DSP_AMD_SHORT_VT=$vt python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c5/c5_vt${vt}_$rep.json 2> gpurun_out/c5/c5_vt${vt}_$rep.err
  python -c "
import json; d=json.load(open('gpurun_out/c5/c5_vt${vt}_$rep.json')); print('config 5 VT=$vt rep $rep:', round(d['value']), 'Msamples/s', {k: round(v['avg_ms'],2) for k,v in d['roofline']['kernels'].items()})"
done; done
VERIFY_ARGS="" scripts/r06_verify.sh 2 > gpurun_out/c5/verify_noslack.txt 2>&1; cat gpurun_out/c5/verify_noslack.txt | cut -c1-200
export DSP_AMD_TESTS_ONE_PROCESS=1
DSP_AMD_GUARD=5 timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fallbacks.py --deselect tests/test_gpu_dropin.py --deselect tests/test_gpu_endpoints.py::test_bench_launches_its_own_ranks --deselect tests/test_gpu_endpoints.py::test_bench_as_a_scale_run_launches_it_eight_ranks_at_the_headline > gpurun_out/c5/pytest_poison.log 2>&1; echo "poison rc $?: $(tail -1 gpurun_out/c5/pytest_poison.log | cut -c1-200)"; grep "^FAILED" gpurun_out/c5/pytest_poison.log | cut -c1-200 | head -20
