# run() time of the LADSPA frontend (reference ladspa_dsp.c linked against libdsp_amd.so) against the all-CPU reference build,
# stereo 10-biquad EQ, at the block sizes LADSPA hosts use
set -e
d=$(mktemp -d)
python - "$d" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import test_gpu_ladspa as T
T.write_configs(sys.argv[1])
np.save(sys.argv[1] + "/in.npy", np.random.default_rng(3).uniform(-0.5, 0.5, (400000, 2)).astype(np.float32))
PY
for lib in ladspa_dsp_ref.so ladspa_dsp_gpu.so; do
	for kb in 0 32; do
		[ $lib = ladspa_dsp_ref.so ] && [ $kb = 32 ] && continue
		echo "$lib DSP_AMD_PLUGIN_MAPPED_KB=$kb"
		DSP_AMD_PLUGIN_MAPPED_KB=$kb python tests/ladspa_host.py oracle/_ref/$lib $d ladspa_dsp 48000 64,256,1024,4096 $d/in.npy $d/out.npy
	done
done
rm -rf $d
