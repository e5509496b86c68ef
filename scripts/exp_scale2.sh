# per-rank step at the stream counts one rank sees at N = 8, 4, 2, 1 GPUs (strong scaling: 256 streams in all, no data-path collective)
# -> gpurun_out/scale2.json (copied to profiles/r02_per_rank_steps.json)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python - <<'PY'
import json, subprocess, sys
out = {}
for s in (32, 64, 128, 256):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--streams", str(s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    k = {n: v["avg_ms"] * v["launches_per_step"] for n, v in d["roofline"]["kernels"].items()}
    out[s] = {"ms_per_step": d["ms_per_step"], "Msamples_per_s": d["value"], "kernels_ms": k, "plan": d["config"]["plan"]}
    print(s, round(d["ms_per_step"], 3), {n: round(v, 3) for n, v in k.items()})
base = out[256]["ms_per_step"]
est = {f"{256 // s} GPUs": base / out[s]["ms_per_step"] for s in (32, 64, 128, 256)}
print("estimated strong scaling:", {k: round(v, 2) for k, v in est.items()})
json.dump({"what": "per-rank step (983040 frames) on ONE MI355X at the stream counts a rank holds when 256 streams are sharded over 8 / 4 / 2 / 1 GPUs; "
                   "estimate = step(256 streams) / step(256 / G streams): no data-path collective, so a rank's step is the job's step",
           "by_streams_per_rank": out, "estimated_speedup": est}, open("gpurun_out/scale2.json", "w"), indent=1)
PY
