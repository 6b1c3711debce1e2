export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04b_pytest.log 2>&1; tail -3 $O/r04b_pytest.log
timeout 600 python tools/step_vs_n.py --json $O/r04b_step_vs_n.json 2>&1 | grep -v amdgpu.ids > $O/r04b_step_vs_n.txt; cat $O/r04b_step_vs_n.txt | cut -c1-400
mkdir -p $O/r04b_prof_small && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04b_prof_small -o s -- python $R/tools/step_vs_n.py --only 10 > $R/$O/r04b_prof_small/run.log 2>&1)
timeout 600 python bench.py --no-extra --no-cpu-baseline --no-pmc 2>/dev/null | cut -c1-1500
