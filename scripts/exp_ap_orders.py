import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import dsp_amd
from oracle_api import RefChain, rms
rng = np.random.Generator(np.random.PCG64(3))
x = rng.uniform(-0.4, 0.4, size=(20000, 2))
for n in (12, 16, 20, 24, 28, 32):
    for d in ("0.3S", "7.77S"):
        chain = f"delay -f{n} {d}"
        try:
            y = dsp_amd.EffectsChain(chain, 48000, 2).process(x, block=4096)
        except ValueError as e:
            print(chain, "refused"); continue
        ref = RefChain(chain, 48000, 2).process(x, block=4096)
        print(chain, y.shape == ref.shape, "rms diff %.2e" % rms(y - ref))
