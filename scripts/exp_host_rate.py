# PCIe-inclusive rate of dspamd_chain_run as a C host sees it: the caller's two buffers allocated and touched once, the C entry point called
# directly (scripts/exp_plugin_rate.py goes through the numpy wrapper, which allocates and copies a fresh output array per call)
import sys, time, ctypes, numpy as np
sys.path.insert(0, '.')
import dsp_amd
B10 = "lowpass 1k 0.707 highshelf 8k 0.7 -3 eq 100 1.0 3 eq 200 1.0 -2 eq 400 2.0 1.5 eq 800 1.0 -1 eq 1600 1.4 2 eq 3200 1.0 -2.5 eq 6400 3.0 1 highpass 20 0.707"
for ch, block in ((8, 2048), (8, 65536), (8, 1 << 18), (8, 1 << 20), (8, 1 << 22), (64, 1 << 17), (2, 1 << 20)):
    ec = dsp_amd.EffectsChain(B10, 48000, ch)
    x = np.ascontiguousarray(np.random.default_rng(1).uniform(-0.5, 0.5, size=(block, ch)))
    out = np.zeros((block, ch))
    run = lambda: ec.L.dspamd_chain_run(ec.h, x.ctypes.data, block, out.ctypes.data, block)
    assert run() == block and run() == block
    ref = ec.run(x)                                   # (the wrapper's path: same entry point, its own arrays)
    assert run() == block
    n = max(3, int(4e8 / (block * ch)))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): run()
        ts.append((time.perf_counter() - t0) / n)
    dt = sorted(ts)[1]
    print(f"{ch} ch, block {block}: {block * ch / dt / 1e6:.1f} Msamples/s ({dt * 1e3:.3f} ms per block, {block * ch * 8 / dt / 1e9:.2f} GB/s each way)", flush=True)
