export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out
for c in 95 136; do
mkdir -p $O/r04p_rc5_$c && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04p_rc5_$c -o s -- python $R/tools/nb_probe.py $c 50 1.38313 0.0 10 > $R/$O/r04p_rc5_$c/run.log 2>&1)
echo "== $c"; grep -v "^W2\|^E2\|amdgpu.ids" $O/r04p_rc5_$c/run.log | tail -3 | cut -c1-300
python - <<P
import csv
rows=list(csv.DictReader(open("$O/r04p_rc5_$c/s_kernel_stats.csv")))
for r in rows[:8]:
    if 'mdh' in r['Name']: print(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3)
P
done
echo "== stamps 95"; NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 95 50 1.38313 0.0 3 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500
echo "== stamps 136"; NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 136 50 1.38313 0.0 3 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500
