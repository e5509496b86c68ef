for a in "256 8 196608" "128 8 98304" "64 8 131072" "32 8 65536" "256 8 196608 fir_p 65536" "32 8 196608 fir_p 65536"; do timeout 120 python scripts/check_determinism.py $a 2>&1 | tail -1; done
for e in 0; do DSP_AMD_CASCADE_ROWS=$e timeout 120 python scripts/check_determinism.py 32 8 65536 2>&1 | tail -1; done
