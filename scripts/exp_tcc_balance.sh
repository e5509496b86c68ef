#!/bin/bash
# memory-side evidence for the pair padding (W, rings): L2 -> fabric request counters of the FFT kernels with and without it.
# TCC_EA0_RDREQ per channel instance (when the tool reports dimensions), DRAM credit stalls, write stalls.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2tcc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in pad nopad; do
  if [ $cfg = nopad ]; then export DSP_AMD_CONV_WPAD=0 DSP_AMD_CONV_RPAD=0; else unset DSP_AMD_CONV_WPAD DSP_AMD_CONV_RPAD; fi
  rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum --kernel-trace -d $O/$cfg -o pmc --output-format csv -- python $R/bench.py --block 196608 --steps 3 --warmup 1 --no-cpu-baseline > $O/$cfg.json 2> $O/$cfg.err
done
python - $O <<'PY'
import csv, sys, glob, collections, json
O = sys.argv[1]
res = {}
for cfg in ("pad", "nopad"):
    f = glob.glob(f"{O}/{cfg}/**/pmc_counter_collection.csv", recursive=True)
    if not f: print("no csv", cfg); continue
    rows = list(csv.DictReader(open(f[0])))
    print(cfg, "columns:", list(rows[0].keys()))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void dspamd::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[cfg] = {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in d.items()} for k, d in acc.items() if "conv" in k or "cascade" in k}
json.dump(res, open(f"{O}/tcc_summary.json", "w"), indent=1)
for cfg, d in res.items():
    for k, c in d.items():
        print(cfg, k, {n: round(v["mean"]) for n, v in c.items()})
PY
find $O -name "*.csv" -size +8M -delete
