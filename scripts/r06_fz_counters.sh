#!/bin/bash
# usage (GPU box): scripts/r06_fz_counters.sh <tag> [bench args, e.g. --config 3]
# Counters of the fused first pass (and of whatever else the step runs): SQ issue / wait split, LDS conflicts, L1 / L2 traffic and hit rates, occupancy.
# One rocprofv3 --pmc pass per counter set (kernel-trace only), averaged per kernel over the launches of 3 steps -> gpurun_out/<tag>/ctr.json + a table.
tag=${1:-fzc}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "TCC_REQ_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"; do
	i=$((i+1))
	rocprofv3 --pmc $ctrs --kernel-trace -d $O/p$i -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-runs "$@" > $O/p$i.log 2> $O/p$i.err || echo "pass $i failed: $(tail -2 $O/p$i.err | cut -c1-200)"
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
        if "dspamd" not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"].split("(")[0][-44:], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, di), cs in per.items():
        if dur.get(di, 0) < float(__import__("os").environ.get("MIN_NS", "5e5")): continue
        agg[k]["ns"].append(dur[di])
        for c, v in cs.items(): agg[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in a.items()} for k, a in agg.items()}
json.dump(out, open(f"{O}/ctr.json", "w"), indent=1)
for k, a in out.items():
    print(k)
    wc = a.get("SQ_WAVE_CYCLES", 0)
    for c, v in sorted(a.items()):
        print(f"    {c:32s} {v:18.0f}" + (f"   {v / wc:7.3f} of wave-cycles" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""))
PY
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
