#!/bin/bash
mkdir -p gpurun_out/c7
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_ladspa.py tests/test_gpu_soak.py -m gpu -x -q > gpurun_out/c7/pytest_resident.log 2>&1; echo "resident/ladspa/soak rc $?: $(tail -1 gpurun_out/c7/pytest_resident.log | cut -c1-200)"
bash scripts/r06_resident_timing.sh > gpurun_out/c7/resident_timing.txt 2>&1; cat gpurun_out/c7/resident_timing.txt | cut -c1-400
{ echo "== default (request mailbox in device memory)"; bash scripts/exp_ladspa_rate.sh 2>&1 | grep -v "^$"; echo "== DSP_AMD_PLUGIN_MAILBOX=host"; DSP_AMD_PLUGIN_MAILBOX=host bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so DSP_AMD_PLUGIN_MAPPED_KB=32"; echo "== DSP_AMD_PLUGIN_RESIDENT=0"; DSP_AMD_PLUGIN_RESIDENT=0 bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so DSP_AMD_PLUGIN_MAPPED_KB=32"; echo "== crossover config"; bash scripts/exp_ladspa_rate_xover.sh 2>&1; } > gpurun_out/c7/ladspa_rate.txt 2>&1
grep -o "^==.*\|^ladspa_dsp_[a-z]*.so.*\|run_seconds.*" gpurun_out/c7/ladspa_rate.txt | cut -c1-200
