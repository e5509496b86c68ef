"""which stages speak the wire formats for a few plans (bits of dspamd_batch_wire_fused per call)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dsp_amd
EQ10 = " ".join(f"eq {f} 1.2 {g}" for f, g in zip((60, 120, 250, 500, 1000, 2000, 4000, 8000, 12000, 16000), (1.5, -2, 1, -1, 2, -1.5, 1, -2, 1.5, -1)))
h = np.random.default_rng(1).standard_normal(700) / 50
np.asarray(h, dtype="<f8").tofile("/tmp/h700.raw")
for chain, S, C in ((f"{EQ10} fir -t pcm -e double -c 1 /tmp/h700.raw", 24, 3), (f"gain 6 fir_p -t pcm -e double -c 1 /tmp/h700.raw", 24, 5),
                    (f"gain 8 {EQ10}", 64, 8), (f"{EQ10} fir_p -t pcm -e double -c 1 /tmp/h700.raw", 128, 8)):
    b = dsp_amd.BatchChain(chain, 48000, C, S, 4096)
    bits = []
    for n in (4096, 1100, 3000):
        x = torch.zeros((S, n, C), dtype=torch.int16, device="cuda")
        b.run_wire(x, "s16", "s16", 16)
        bits.append(b.wire_fused())
    print(bits, b.plan())
