# cascade kernels: the channel groups of a stream co-scheduled on one XCD (default) against the plain block order
BQ=$(python -c "import bench; print(bench.BIQUADS)")
one() { python bench.py "$@" --no-cpu-baseline --no-side-runs --steps 10 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],2), {k: round(v['avg_ms'],2) for k,v in r['roofline']['kernels'].items()})"; }
for m in 1 0; do
  echo "== DSP_AMD_CASCADE_XCDMAP=$m: sections alone (interleaved output) / headline / headline with cascade_fast / 64 streams sections alone with cascade_wave"
  DSP_AMD_CASCADE_XCDMAP=$m one --chain "$BQ"
  DSP_AMD_CASCADE_XCDMAP=$m one
  DSP_AMD_CASCADE_XCDMAP=$m DSP_AMD_CASCADE_ROWS=0 one
  DSP_AMD_CASCADE_XCDMAP=$m DSP_AMD_CASCADE_ROWS=0 one --chain "$BQ" --streams 64
done
