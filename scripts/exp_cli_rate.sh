# the drop-in itself: the unmodified reference CLI (null output, raw PCM input from /dev/shm) with its own effects (dsp_ref)
# and linked against libdsp_amd.so (dsp_gpu), 8 ch x 10 biquads, at the default block size and at larger ones
B10="lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
python - <<'PY'
import numpy as np
np.random.default_rng(1).uniform(-0.5, 0.5, size=(48000 * 600, 8)).astype('<f8').tofile('/dev/shm/in8.raw')
PY
for exe in dsp_ref dsp_gpu; do for b in 2048 65536; do
  for pin in d 0 m; do   # the default (mapped staging up to 32 KB, else pageable copies) / pageable copies only / mapped staging forced up to 256 KB
  [ $exe = dsp_ref ] && [ $pin != d ] && continue
  kb=32; [ $pin = 0 ] && kb=0; [ $pin = m ] && kb=256
  s=$(date +%s.%N)
  DSP_AMD_PLUGIN_MAPPED_KB=$kb oracle/_ref/$exe -q -b $b -t pcm -e double -r 48k -c 8 /dev/shm/in8.raw -o -t null null $B10
  e=$(date +%s.%N)
  python -c "print('$exe block $b staging $pin: %.2f s -> %.0f Msamples/s' % ($e - $s, 48000 * 600 * 8 / ($e - $s) / 1e6))"
  done
done; done
rm -f /dev/shm/in8.raw
