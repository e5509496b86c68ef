#!/bin/bash
# where the time of a 64-frame block through the resident wave goes: the LADSPA frontend on a build of the library with -DRES_TIMING
# (scripts/exp_libs/libdsp_amd_timing.so: kernels_resident.hip and plugin.cpp compiled with the flag), 64-frame blocks only, both mailbox placements
set -e
d=$(mktemp -d)
python - "$d" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import test_gpu_ladspa as T
T.write_configs(sys.argv[1])
np.save(sys.argv[1] + "/in.npy", np.random.default_rng(3).uniform(-0.5, 0.5, (200000, 2)).astype(np.float32))
PY
cp dsp_amd/libdsp_amd.so $d/keep.so
cp scripts/exp_libs/libdsp_amd_timing.so dsp_amd/libdsp_amd.so
for mb in device host; do for b in 64 128; do
  echo "== mailbox $mb, blocks of $b"
  DSP_AMD_PLUGIN_MAILBOX=$mb python tests/ladspa_host.py oracle/_ref/ladspa_dsp_gpu.so $d ladspa_dsp 48000 $b $d/in.npy $d/out.npy 2>&1 | grep -o "run_seconds.*\|resident timing.*"
done; done
cp $d/keep.so dsp_amd/libdsp_amd.so
rm -rf $d
