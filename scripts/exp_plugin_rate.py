# PCIe-inclusive rate of the host-buffer path (dspamd_chain_run: H2D + device segment + D2H per block)
import sys, time, numpy as np
sys.path.insert(0, '.')
import dsp_amd
B10 = "lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
for ch, block in ((2, 64), (2, 1024), (8, 2048), (8, 65536), (8, 1 << 20), (64, 1 << 17)):
    ec = dsp_amd.EffectsChain(B10, 48000, ch)
    x = np.random.default_rng(1).uniform(-0.5, 0.5, size=(block, ch))
    ec.run(x); ec.run(x)
    n = max(3, int(2e8 / (block * ch)))
    t0 = time.perf_counter()
    for _ in range(n): ec.run(x)
    dt = time.perf_counter() - t0
    print(f"{ch} ch, block {block}: {n * block * ch / dt / 1e6:.1f} Msamples/s ({dt / n * 1e3:.3f} ms per block, {n * block * ch * 16 / dt / 1e9:.2f} GB/s over PCIe both ways)")
