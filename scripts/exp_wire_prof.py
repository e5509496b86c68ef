"""per-kernel times of the headline step run from wire format to wire format (dspamd_batch_run_wire) for a few format pairs"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dsp_amd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import BIQUADS, make_filter
L = dsp_amd.load_library()
S, C, block, pad = 256, 8, 983040, 68
d = "/tmp/wire_prof"; os.makedirs(d, exist_ok=True)
np.asarray(make_filter(65536), dtype="<f8").tofile(d + "/filt.raw")
chain = BIQUADS + " fir_p -t pcm -e double -c 1 filt.raw"
DT = {"s16": torch.int16, "s24": torch.int32, "s32": torch.int32, "float": torch.float32, "double": torch.float64}
cases = [("double", "double", 0), ("s16", "s16", 16), ("s16", "s16", 0), ("float", "float", 0), ("s32", "s32", 0), ("double", "s16", 16), ("s16", "double", 0)]
NOSTATS = "nostats" in sys.argv
if len(sys.argv) > 1:
    cases = [tuple(a.split(",")[:2]) + (int(a.split(",")[2]),) for a in sys.argv[1:] if "," in a]
NOSTATS = "nostats" in sys.argv
for fi, fo, prec in cases:
    b = dsp_amd.BatchChain(chain, 48000, C, S, block, directory=d)
    x = torch.zeros((S, block + pad, C), dtype=DT[fi], device="cuda")
    if fi in ("float", "double"):
        x[:, :block, :] = torch.rand((1, block, C), device="cuda", dtype=torch.float32).to(DT[fi]) - 0.5
    else:
        x[:, :block, :] = torch.randint(-20000, 20000, (1, block, C), device="cuda", dtype=torch.int32).to(DT[fi])
    o = torch.empty((S, block + pad, C), dtype=DT[fo], device="cuda")
    st = None if NOSTATS else torch.zeros((S, 2), dtype=torch.float64, device="cuda")
    for _ in range(2):
        b.run_wire(x[:, :block, :], fi, fo, prec, st, o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.run_wire(x[:, :block, :], fi, fo, prec, st, o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    L.dspamd_profile_enable(1)
    for _ in range(3):
        b.run_wire(x[:, :block, :], fi, fo, prec, st, o)
    prof = {l.split()[0]: float(l.split()[1]) / int(l.split()[2]) for l in L.dspamd_profile_collect().decode().splitlines()}
    L.dspamd_profile_enable(0)
    print(f"{fi:>6} -> {fo:<6} dither {prec:2d}: {dt*1e3:6.2f} ms/step bits {b.wire_fused()}  " + "  ".join(f"{k} {v:.2f}" for k, v in prof.items()), flush=True)
    del b, x, o
    torch.cuda.empty_cache()
