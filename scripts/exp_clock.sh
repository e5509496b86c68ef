#!/bin/bash
# usage (GPU box): bash scripts/exp_clock.sh <tag>
# Effective shader clock per kernel = GRBM_GUI_ACTIVE / kernel wall time (MI355X_MICROARCH.md, DVFS section), with the SQ
# issue counters beside it, for (i) the headline step's kernels and (ii) a pure fp64-FMA stream (ubench/fp64bench) at the
# same occupancies -- the counter-backed answer to "what bounds cascade_rows<4>".
tag=${1:-clock}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/scripts/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 fp64bench.hip -o fp64bench 2>$O/build.err
cd /tmp && export TMPDIR=/tmp
run_pass() {  # name, counters, command...
	name=$1; ctrs=$2; shift 2
	rocprofv3 --pmc $ctrs --kernel-trace -d $O/$name -o pmc --output-format csv -- "$@" > $O/$name.log 2> $O/$name.err
}
run_pass bench_a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline
run_pass bench_b "SQ_INST_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline
run_pass bench_c "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64" python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline
run_pass fp64_a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" $R/scripts/ubench/fp64bench
run_pass fp64_b "SQ_INST_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" $R/scripts/ubench/fp64bench
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
out = {}
for name in ("bench_a", "bench_b", "bench_c", "fp64_a", "fp64_b"):
    cf = glob.glob(f"{O}/{name}/**/pmc_counter_collection.csv", recursive=True)
    kf = glob.glob(f"{O}/{name}/**/pmc_kernel_trace.csv", recursive=True)
    if not cf or not kf:
        out[name] = {"error": "no csv", "files": glob.glob(f"{O}/{name}/**/*", recursive=True)[:8]}
        continue
    dur = {}
    for r in csv.DictReader(open(kf[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cf[0])):
        k = r["Kernel_Name"][:70]
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    order = []
    for (k, d), cs in per.items():
        key = k if name.startswith("bench") else f"{k} #{d}"
        if key not in order: order.append(key)
        agg[key]["ns"].append(dur.get(d, (None, float("nan")))[1])
        for c, v in cs.items(): agg[key][c].append(sum(v))
    res = {}
    for key in order:
        if name.startswith("bench") and "dspamd" not in key: continue
        a = {c: sum(v) / len(v) for c, v in agg[key].items()}
        a["launches"] = len(agg[key]["ns"])
        if "GRBM_GUI_ACTIVE" in a:
            a["GUI_ACTIVE_per_ns"] = a["GRBM_GUI_ACTIVE"] / a["ns"]
            a["GHz_if_summed_over_8_XCD"] = a["GRBM_GUI_ACTIVE"] / 8 / a["ns"]
        if "SQ_ACTIVE_INST_VALU" in a and "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"]:
            a["valu_active_per_wave_cycle"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]
        res[key] = a
    out[name] = res
json.dump(out, open(f"{O}/clock.json", "w"), indent=1)
for n, r in out.items():
    print("==", n)
    for k, a in r.items(): print("  ", k[:60], {c: (round(v, 3) if isinstance(v, float) else v) for c, v in a.items()} if isinstance(a, dict) else a)
PY
find $O -name "*.csv" -size +20M -delete
