#!/usr/bin/env python3
"""usage: make_blocksweep.py <gpurun_out/TAG> <profiles/NAME.json> [default-step bench json]
Collects the bench lines of scripts/exp_blocksweep.sh (one per call size) into one file: Msamples/s, ms per call, per-kernel ms per call
(HIP events inside bench.py) and the plan of each call size."""
import glob, json, os, sys

src, dst = sys.argv[1], sys.argv[2]
out = {"what": "headline chain (256 x 8 ch, biquad x10 + fir_p(65536)) at call sizes from 256 frames to the default step; bench.py --block B, "
               "MI355X, round 3 (HEAD of the round); per-kernel ms per call from HIP events", "by_block": {}}
files = sorted(glob.glob(os.path.join(src, "block_*.json")), key=lambda f: int(os.path.basename(f)[6:-5]))
if len(sys.argv) > 3:
    files.append(sys.argv[3])
for f in files:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    k = {n: v["avg_ms"] * v["launches_per_step"] for n, v in d["roofline"]["kernels"].items()}
    out["by_block"][str(d["config"]["block_frames"])] = {"Msamples_per_s": d["value"], "ms_per_call": d["ms_per_step"], "kernels_ms": k, "plan": d["config"]["plan"]}
json.dump(out, open(dst, "w"), indent=1)
for b, v in out["by_block"].items():
    print(b, round(v["Msamples_per_s"]), round(v["ms_per_call"], 4))
