# the GPU suite once per fallback path (each switch removes one fast path; results must not change).  Tests that assert on
# the plan text of the path that was switched off are left out.
SKIP="not chunked and not slab_direct and not registered_host and not chained_convolvers_feed and not headline_workload_full_size and not bench_default_configuration and not convolver_configs_full_size and not small_calls"
# round 2 switches: the one-shot K2, no small-call regime, a barrier per cascade step, no pair padding, no nt hints, wire formats always as passes of their own
# (ROUND=2 runs only those)
R1="DSP_AMD_CASCADE_ROWS=0|DSP_AMD_CASCADE_ROWS=0 DSP_AMD_CASCADE_WAVE=0|DSP_AMD_CASCADE_CHUNKS=0|DSP_AMD_CONV_NO_DIRECT=1|DSP_AMD_PLUGIN_MAPPED_KB=0|DSP_AMD_NO_LTI_MERGE=1|DSP_AMD_NO_FEED=1"
R2="DSP_AMD_ROW_PIPE=0|DSP_AMD_CONV_FDL=0|DSP_AMD_CASCADE_P2P=0|DSP_AMD_CONV_WPAD=0 DSP_AMD_CONV_RPAD=0|DSP_AMD_CONV_NT=0|DSP_AMD_NO_WIRE_FUSION=1|DSP_AMD_CASCADE_XCDMAP=0"
LIST="$R1|$R2"; [ "$ROUND" = 2 ] && LIST="$R2"
IFS='|'
for e in $LIST; do
	unset IFS
	[ -n "$ONLY" ] && [ "$e" != "$ONLY" ] && continue
	echo "== $e"
	env $e timeout 600 python -m pytest tests -q -m gpu -k "$SKIP" 2>&1 | tail -4 | cut -c1-300
done
