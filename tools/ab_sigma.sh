#!/bin/bash
out=${1:-gpurun_out/ab_sigma.txt}
{
for s in 0.05 0.2; do
for t in 1 0; do
  echo "== sigma $s MDH_LANE_TK8=$t"; MDH_LANE_TK8=$t python tools/nb_probe.py 136 16 0.854 $s 10 2>&1 | grep -v amdgpu.ids
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sigma -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/nb_probe.py 136 16 0.854 0.05 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_sigma/**/p_kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
} > $out 2>&1
tail -40 $out
