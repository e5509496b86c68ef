#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from a scripts/profile_round.sh run: HBM bytes per launch of the bench's kernels.

usage: make_traffic.py gpurun_out/<tag> <tag>
FETCH_SIZE / WRITE_SIZE are reported in KiB.  Corrections (MI355X_MICROARCH.md "HBM", checked in the same run against
copy_probe_kernel = 1 GiB read + 1 GiB write): FETCH_SIZE under-counts wide contiguous 16 B/lane loads by 2x on gfx950
(factor taken from the probe).  The cascade reads half a 64-byte frame per lane group: as long as the two channel groups of a
stream ran a dispatch wave apart (up to r02d) those 32-byte requests reached HBM as they were and the counter took them 1:1
(14.7 GiB raw for 15 GiB read); since the groups are co-scheduled on one XCD (r02e) the L2 merges them into whole-line fetches,
which the counter under-counts like any wide load: 7.35 GiB raw for the same 15 GiB -- the probe's factor applies (argument
`cascade_factor`, default: the probe's).  conv_col_fwd's average is diluted by the
one-pair filter-preparation launch: scaled by launches / (launches - 1)."""
import json, sys, os

src, tag = sys.argv[1], sys.argv[2]
cascade_factor = float(sys.argv[3]) if len(sys.argv) > 3 else None
raw = json.load(open(os.path.join(src, "pmc_hbm.json")))
bench = json.load(open(os.path.join(src, "bench.json")))
cfg = bench["config"]
S, C, B, T = cfg["streams"], cfg["channels"], cfg["block_frames"], cfg["taps"]
samples = S * C * B
N = 1
while N < B + T - 1: N *= 2
pairs = S * C // 2

def find(sub):
    for k, v in raw.items():
        if sub in k: return k, v
    return None, None

probe_k, probe = find("copy_probe_kernel")
fetch_factor = (1 << 20) / probe["FETCH_SIZE"]["per_launch_raw"] if probe else 2.0
kernels = {}
spec = {   # bench name -> (rocprof substring, fetch factor, designed bytes)
    "cascade_rows": ("cascade_rows", cascade_factor if cascade_factor else fetch_factor, samples * 16),
    "cascade_fast": ("cascade_fast", 1.0, samples * 16),
    "cascade_wave": ("cascade_wave", 1.0, samples * 16),
    "conv_col_fwd": ("conv_col_fwd", fetch_factor, pairs * N * 32),
    # round 4, the cascade fused into the first pass: the prepass reads the call's frames once (its end states are a few MB); the fused
    # first pass reads them again plus the history rows from the rings, writes W and the next window's history rows
    "fused_prepass": ("fused_prepass", fetch_factor, samples * 8),
    "fused_col_fwd": ("fused_col_fwd", fetch_factor, samples * 8 + pairs * (N - B) * 16 * 2 + pairs * N * 16),
    "cascade_chunk_carry": ("cascade_chunk_carry", 1.0, 0),
    "conv_row": ("conv_row_duo" if find("conv_row_duo")[0] else "conv_row_pipe", fetch_factor, pairs * N * 32),        # (16 B/lane buffer loads / LDS-DMA: same under-count, MI355X_MICROARCH.md)
    "conv_col_inv": ("conv_col_inv", fetch_factor, pairs * N * 16 + samples * 8),
}
fused = find("fused_col_fwd")[0] is not None
for name, (sub, ff, designed) in spec.items():
    if fused and name in ("cascade_rows", "cascade_fast", "cascade_wave", "conv_col_fwd"): continue   # (the drain's and the filter preparation's launches, not the step's)
    k, v = find(sub)
    if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v: continue
    dil = 1.0
    if name == "conv_col_fwd" and not find("fused_col_fwd")[0]:
        n = v["FETCH_SIZE"]["launches"]
        dil = n / (n - 1.0) if n > 1 else 1.0
    fr = v["FETCH_SIZE"]["per_launch_raw"] * 1024 * dil
    wr = v["WRITE_SIZE"]["per_launch_raw"] * 1024 * dil
    kernels[name] = {"rocprof_name": k.split("(")[0].replace("void ", ""), "fetch_raw": fr, "fetch_corrected": fr * ff, "write": wr,
                     "traffic": fr * ff + wr, "algorithmic_bytes_of_kernel": float(designed)}
out = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`, {tag}; scripts/profile_round.sh + scripts/make_traffic.py",
    "units": "bytes per launch",
    "calibration": f"copy_probe_kernel (1 GiB read + 1 GiB write, 16 B/lane, contiguous) in the same run: FETCH_SIZE factor {fetch_factor:.3f}, WRITE_SIZE {probe['WRITE_SIZE']['per_launch_raw'] / (1 << 20):.3f} GiB" if probe else "no probe",
    "kernels": kernels,
    "workload_key": f"{S}x{C}x{B}x{T}",
}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
for k, v in kernels.items():
    print(f"  {k:14s} fetch {v['fetch_corrected'] / 2**30:6.3f} GiB  write {v['write'] / 2**30:6.3f} GiB  total {v['traffic'] / 2**30:6.3f} GiB  designed {v['algorithmic_bytes_of_kernel'] / 2**30:6.3f} GiB")
