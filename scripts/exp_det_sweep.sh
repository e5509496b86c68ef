# bit-for-bit repeatability across the kernel families (scripts/check_determinism.py)
c() { CHAIN="$1" python scripts/check_determinism.py $2 $3 $4 2>&1 | tail -1 | sed "s|^|[$1] |" | cut -c1-220; }
c "resample 44.1k" 64 8 65536
c "resample 96k" 64 8 65536
c "resample 0.9 32k" 16 2 50000
c "lowpass -r 1k 0.707 highpass -r 100 0.707" 64 2 65536
c "gain -3 remix 0,1 2 . 1,2,3 :0 delay 37S : lowpass 2k 0.7" 64 4 30000
c "st2ms :0 lowpass 3k 0.7 : ms2st crossfeed 700 4.5 delay -f 0.3S" 128 2 40000
c "hilbert -p 1023 :1 gain -2 : resample 44.1k" 32 2 60000
c "hilbert 4095 fir coefs:0.5,0.25,-0.125,0.0625,0.03,0.01,0.5,0.25,-0.125,0.0625,0.03,0.01,0.5,0.25,-0.125,0.0625,0.03,0.01" 16 3 30000
c ":0,1 lowpass 1k 0.7 :2 eq 300 1 3 : highshelf 5k 0.7 -2" 200 8 20000
c "gain -2 add 0.001 mult 0.9" 300 8 8192
