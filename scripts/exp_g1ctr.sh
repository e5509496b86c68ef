#!/bin/bash
# cascade_rows<1> at 32 streams x 8 ch (8-byte accesses 64 bytes apart) against 128 x 2 ch (16 bytes apart): what waits for what
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g1ctr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in "32 8" "128 2"; do set -- $shape
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"; do
	i=$((i+1))
	timeout 300 rocprofv3 --pmc $ctrs --kernel-trace -d $O/s$1_p$i -o pmc --output-format csv -- python $R/bench.py --streams $1 --channels $2 --steps 3 --warmup 1 --no-cpu-baseline --no-side-runs > $O/s$1_p$i.log 2> $O/s$1_p$i.err || tail -3 $O/s$1_p$i.err
done
done
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
out = {}
for s in ("32", "128"):
    agg = collections.defaultdict(list)
    for d in sorted(glob.glob(f"{O}/s{s}_p*/")):
        cf = glob.glob(d + "**/pmc_counter_collection.csv", recursive=True); kf = glob.glob(d + "**/pmc_kernel_trace.csv", recursive=True)
        if not cf or not kf: print("no csv in", d); continue
        dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kf[0]))}
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(cf[0])):
            if "cascade_rows" not in r["Kernel_Name"]: continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for di, cs in per.items():
            if dur.get(di, 0) < 5e5: continue
            agg["ns"].append(dur[di])
            for c, v in cs.items(): agg[c].append(v)
    out[s] = {c: sum(v) / len(v) for c, v in agg.items()}
json.dump(out, open(f"{O}/ctr.json", "w"), indent=1)
keys = sorted(set(out["32"]) | set(out["128"]))
for c in keys: print(f"{c:40s} {out['32'].get(c, 0):16.0f} {out['128'].get(c, 0):16.0f}")
PY
find $O -name "*.csv" -size +20M -delete
