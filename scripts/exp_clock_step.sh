#!/bin/bash
# usage (GPU box): bash scripts/exp_clock_step.sh <tag> [env...]   clock and VALU share of every kernel of the step: GRBM_GUI_ACTIVE / duration
# (summed over the 8 XCDs), SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)
tag=${1:-clock}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/p1 -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-runs > $O/p1.log 2> $O/p1.err
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
cf = glob.glob(O + "/p1/**/pmc_counter_collection.csv", recursive=True); kf = glob.glob(O + "/p1/**/pmc_kernel_trace.csv", recursive=True)
dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kf[0]))}
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cf[0])):
    per[(r["Kernel_Name"].split("(")[0][-44:], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, di), cs in per.items():
    if dur.get(di, 0) < 1e6: continue
    agg[k]["ns"].append(dur[di])
    for c, v in cs.items(): agg[k][c].append(v)
out = {}
for k, a in agg.items():
    m = {c: sum(v) / len(v) for c, v in a.items()}
    m["clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / m["ns"]
    m["valu_busy"] = m["SQ_ACTIVE_INST_VALU"] * 4 / (m["SQ_BUSY_CYCLES"] / 32 * 1024) if m.get("SQ_BUSY_CYCLES") else None
    out[k] = m
    print(f"{k:46s} {m['ns'] / 1e6:7.3f} ms  clock {m['clock_GHz']:.3f} GHz  VALU busy {m['valu_busy']:.3f}  n={len(a['ns'])}")
json.dump(out, open(f"{O}/clock.json", "w"), indent=1)
PY
find $O -name "*.csv" -size +20M -delete
