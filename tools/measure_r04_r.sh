export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out
for c in 63 100; do
mkdir -p $O/r04r_$c && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04r_$c -o s -- python $R/tools/nb_probe.py $c 16 0.854 0.0 20 > $R/$O/r04r_$c/run.log 2>&1)
echo "== $c"; grep -v "^W2\|^E2\|amdgpu.ids" $O/r04r_$c/run.log | tail -3 | cut -c1-200
python - <<P
import csv
rows=list(csv.DictReader(open("$O/r04r_$c/s_kernel_stats.csv")))
for r in rows[:9]:
    if 'mdh' in r['Name']: print(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3)
P
done
