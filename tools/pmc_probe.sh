#!/bin/bash
# usage: tools/pmc_probe.sh <tag> <counter list...>   one rocprofv3 PMC pass over tools/nb_probe.py (csv under gpurun_out/pmc_<tag>)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/tools/nb_probe.py ${PROBE_ARGS:-136 16 0.854 0 2} > $R/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag rc=$?"
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_$tag/**/p_counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "neighbor" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
    for k in acc:
        print(k, "dispatches", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
