# usage: prof_pmc.sh <outdir> <counters...> -- <bench args...>
# one rocprofv3 --pmc pass (kernel-trace only), summarised per kernel
out=$1; shift
ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace -d $out -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/bench.log 2> $out/err.log
python - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + '/**/pmc_counter_collection.csv', recursive=True)
if not f: print("no counter file", glob.glob(out + '/**/*', recursive=True)[:10]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    if 'dspamd' not in k: continue
    print(k)
    for c, v in d.items(): print(f"    {c:28s} {v / cnt[(k, c)]:16.1f} per launch ({cnt[(k, c)]} launches)")
PY
