mkdir -p gpurun_out/e1
for mb in 0 32 64 96 128 192 256; do
  echo "chunk_mb=$mb"; DSP_AMD_CONV_CHUNK_MB=$mb python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print(d['value'], d['ms_per_step'], {k:round(v['avg_ms']*v['launches_per_step'],3) for k,v in r.items()})"
done
