#!/bin/bash
# round 6, call 4: the suite (a process per module) on the mailbox wave and the trimmed kernel set; LADSPA block times; the round's profile set;
# counters of the fused first pass (config 3's pass-through instance) and of the one-trip convolver (config 5)
mkdir -p gpurun_out/c4
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c4/pytest.log 2>&1; echo "pytest rc $?: $(tail -3 gpurun_out/c4/pytest.log | cut -c1-300)"
{ echo "== default (request mailbox in device memory)"; bash scripts/exp_ladspa_rate.sh 2>&1 | grep -v "^$"; echo "== DSP_AMD_PLUGIN_MAILBOX=host"; DSP_AMD_PLUGIN_MAILBOX=host bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so DSP_AMD_PLUGIN_MAPPED_KB=32"; echo "== DSP_AMD_PLUGIN_RESIDENT=0"; DSP_AMD_PLUGIN_RESIDENT=0 bash scripts/exp_ladspa_rate.sh 2>&1 | grep -A1 "gpu.so DSP_AMD_PLUGIN_MAPPED_KB=32"; echo "== crossover config"; bash scripts/exp_ladspa_rate_xover.sh 2>&1; } > gpurun_out/c4/ladspa_rate.txt 2>&1
echo "ladspa: $(grep -c run_seconds gpurun_out/c4/ladspa_rate.txt) lines"; head -8 gpurun_out/c4/ladspa_rate.txt | cut -c1-400
bash scripts/exp_cli_rate.sh > gpurun_out/c4/cli_rate.txt 2>&1; cat gpurun_out/c4/cli_rate.txt
bash scripts/profile_round.sh r06a > gpurun_out/c4/profile_round.log 2>&1; tail -12 gpurun_out/c4/profile_round.log | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/r06a/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})
for k,v in d.get('side_runs',{}).get('other_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('kernels'))
"
bash scripts/r06_fz_counters.sh r06_fz_c3 --config 3 > gpurun_out/c4/fz_c3.txt 2>&1; tail -5 gpurun_out/c4/fz_c3.txt | cut -c1-200
bash scripts/r06_fz_counters.sh r06_short_c5 --config 5 > gpurun_out/c4/short_c5.txt 2>&1; tail -5 gpurun_out/c4/short_c5.txt | cut -c1-200
