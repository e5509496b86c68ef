# run() time of the LADSPA frontend's 2 -> 4 crossover config (remix + per-band sections: tests/test_gpu_ladspa.py config_xover), CPU build against the
# library with and without the resident small-block wave
set -e
d=$(mktemp -d)
python - "$d" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import test_gpu_ladspa as T
T.write_configs(sys.argv[1])
np.save(sys.argv[1] + "/in.npy", np.random.default_rng(3).uniform(-0.5, 0.5, (200000, 2)).astype(np.float32))
PY
echo "ladspa_dsp_ref.so"; python tests/ladspa_host.py oracle/_ref/ladspa_dsp_ref.so $d ladspa_dsp:xover 48000 64,128,256 $d/in.npy $d/out.npy | grep -o "run_seconds.*"
echo "ladspa_dsp_gpu.so"; python tests/ladspa_host.py oracle/_ref/ladspa_dsp_gpu.so $d ladspa_dsp:xover 48000 64,128,256 $d/in.npy $d/out.npy | grep -o "run_seconds.*"
echo "ladspa_dsp_gpu.so DSP_AMD_PLUGIN_RESIDENT=0"; DSP_AMD_PLUGIN_RESIDENT=0 python tests/ladspa_host.py oracle/_ref/ladspa_dsp_gpu.so $d ladspa_dsp:xover 48000 64,128,256 $d/in.npy $d/out.npy | grep -o "run_seconds.*"
rm -rf $d
