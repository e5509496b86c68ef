# when does the chunked cascade (kernels_chunk.hip) pay?  10 biquads, few channels, call sizes from 8192 frames up
B10="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --chain "$B10" "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['kernels']; print('   ', round(d['ms_per_step']*1000,1), 'us', {k:round(v['avg_ms']*v['launches_per_step']*1000,1) for k,v in r.items()})"; }
for cfg in "1 2 8192" "1 2 16384" "1 2 65536" "1 8 8192" "1 8 32768" "4 8 16384" "32 8 8192" "32 8 65536" "64 8 32768"; do
	set -- $cfg
	for k in 0 -1; do echo "S=$1 C=$2 frames=$3 chunks=$k"; DSP_AMD_CASCADE_CHUNKS=$k run --streams $1 --channels $2 --block $3; done
done
