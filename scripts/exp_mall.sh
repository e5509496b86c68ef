# does W stay in the 256 MiB Infinity Cache between K1 / K2 / K3 when the batch is launched in chunks of a few streams?
# config 3 (fir_p alone: K1, K2, K3 only); libraries abso/base.so (K2 with nt loads / stores) and abso/nt0.so (without)
run() { # lib, env...
  lib=$1; shift
  cp abso/$lib.so dsp_amd/libdsp_amd.so
  env "$@" python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['roofline']['kernels']
print('$lib $*', round(d['ms_per_step'], 3), {n: round(v['avg_ms'] * v['launches_per_step'], 3) for n, v in k.items()})"
}
run base X=0
run base DSP_AMD_CONV_CHUNK_MB=200
run base DSP_AMD_CONV_CHUNK_MB=200 DSP_AMD_CONV_NT=0
run nt0 DSP_AMD_CONV_CHUNK_MB=200 DSP_AMD_CONV_NT=0
run nt0 DSP_AMD_CONV_CHUNK_MB=130 DSP_AMD_CONV_NT=0
run nt0 DSP_AMD_CONV_CHUNK_MB=70 DSP_AMD_CONV_NT=0
run nt0 DSP_AMD_CONV_NT=0
run nt0 DSP_AMD_CONV_CHUNK_MB=200 DSP_AMD_CONV_NT=0 DSP_AMD_CONV_SUBSTREAMS=2
