#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counter list>" <command...>
# one rocprofv3 PMC pass over any command; per-kernel averages of every mdh:: kernel -> gpurun_out/pmc_<tag>.json
tag=$1; counters=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag rc=$?"
python - <<PY
import csv, glob, collections, json, re
f = glob.glob("$R/gpurun_out/pmc_$tag/**/p_counter_collection.csv", recursive=True)
out = {}
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if "mdh::" not in name: continue
        k = re.sub(r"\(.*", "", name).replace("void ", "")[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
    for k in sorted(acc):
        out[k] = {"dispatches": cnt[k], **{c: v / cnt[k] for c, v in acc[k].items()}}
        print(k, out[k])
json.dump(out, open("$R/gpurun_out/pmc_$tag.json", "w"), indent=1)
PY
