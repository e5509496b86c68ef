#!/bin/bash
# usage (GPU box): bash scripts/exp_fzpmc.sh <tag>   HBM bytes of the step's kernels (FETCH_SIZE / WRITE_SIZE in separate passes)
tag=${1:-fzpmc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-runs > $O/pmc_$c.json 2> $O/pmc_$c.err
done
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{O}/pmc_{c}/**/pmc_counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    for k in acc: res[k][c] = {"launches": cnt[k], "per_launch_raw": acc[k] / cnt[k]}
json.dump(res, open(f"{O}/pmc_hbm.json", "w"), indent=1)
for k, d in res.items():
    if "dspamd" in k: print(k.split("(")[0][-50:], {c: round(v["per_launch_raw"]) for c, v in d.items()}, {c: v["launches"] for c, v in d.items()})
PY
find $O -name "*.csv" -size +20M -delete
