#!/bin/bash
# usage (GPU box): bash scripts/exp_k2ctr.sh <tag> [env...]   SQ / LDS counters of the step's kernels (what the fused kernels wait for)
tag=${1:-fzctr}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
	i=$((i+1))
	env "$@" rocprofv3 --pmc $ctrs --kernel-trace -d $O/p$i -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-runs > $O/p$i.log 2> $O/p$i.err
done
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{O}/p*/")):
    cf = glob.glob(d + "**/pmc_counter_collection.csv", recursive=True); kf = glob.glob(d + "**/pmc_kernel_trace.csv", recursive=True)
    if not cf or not kf: print("no csv in", d); continue
    dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kf[0]))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cf[0])):
        if "fused" not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"].split("(")[0][-40:], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, di), cs in per.items():
        if dur.get(di, 0) < 1e6: continue
        agg[k]["ns"].append(dur[di])
        for c, v in cs.items(): agg[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in a.items()} for k, a in agg.items()}
json.dump(out, open(f"{O}/ctr.json", "w"), indent=1)
for k, a in out.items():
    print(k)
    wc = a.get("SQ_WAVE_CYCLES", 0)
    for c, v in sorted(a.items()):
        print(f"    {c:26s} {v:16.0f}" + (f"   {v / wc:7.3f} of wave-cycles" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""))
PY
find $O -name "*.csv" -size +20M -delete
