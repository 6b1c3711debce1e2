export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out
for sg in 0.0 0.20 0.5; do
mkdir -p $O/r04t_$sg && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04t_$sg -o s -- python $R/tools/fcna_probe.py 136 $sg 20 > $R/$O/r04t_$sg/run.log 2>&1)
echo "== sigma $sg"; grep -v "^W2\|^E2\|amdgpu.ids" $O/r04t_$sg/run.log | tail -1 | cut -c1-300
python - <<P
import csv
rows=list(csv.DictReader(open("$O/r04t_$sg/s_kernel_stats.csv")))
for r in rows[:12]:
    if 'fcna' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3)
P
done
