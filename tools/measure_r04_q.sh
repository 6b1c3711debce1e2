export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "neighbor or dense or unwrapped or variants" 2>&1 | tail -3
for c in 95 120 136; do echo "== $c"; python tools/nb_probe.py $c 50 1.38313 0.0 10 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200; done
mkdir -p $O/r04q_rc5_95 && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04q_rc5_95 -o s -- python $R/tools/nb_probe.py 95 50 1.38313 0.0 10 > $R/$O/r04q_rc5_95/run.log 2>&1)
python - <<P
import csv
rows=list(csv.DictReader(open("$O/r04q_rc5_95/s_kernel_stats.csv")))
for r in rows[:8]:
    if 'mdh' in r['Name']: print(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3)
P
