#!/bin/bash
# round 5, GPU call 11: the resident small-block wave -- its tests, the LADSPA and drop-in tests with it on (default), then run() times with it on and off
mkdir -p gpurun_out/r05k
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_ladspa.py tests/test_gpu_dropin.py -m gpu -x -q > gpurun_out/r05k/pytest.log 2>&1; echo "pytest resident+ladspa+dropin rc $?"; tail -8 gpurun_out/r05k/pytest.log | cut -c1-300
{
echo "== resident wave on (default)"; timeout 600 bash scripts/exp_ladspa_rate.sh 2>&1 | grep -v "^$" | cut -c1-400
echo "== DSP_AMD_PLUGIN_RESIDENT=0"; DSP_AMD_PLUGIN_RESIDENT=0 timeout 600 bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so" | cut -c1-400
} > gpurun_out/r05k/ladspa_rate.txt 2>&1
cat gpurun_out/r05k/ladspa_rate.txt | grep -o "^==.*\|^ladspa.*\|run_seconds.*"
