# round-4 first measurement pass (one gpurun call): baseline numbers before any kernel change of the round
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04a_pytest.log 2>&1; tail -3 $O/r04a_pytest.log
timeout 600 python bench.py > $O/r04a_bench.json 2> $O/r04a_bench.err; tail -c 400 $O/r04a_bench.json
timeout 600 python tools/step_vs_n.py --json $O/r04a_step_vs_n.json 2>&1 | grep -v amdgpu.ids > $O/r04a_step_vs_n.txt; tail -12 $O/r04a_step_vs_n.txt
mkdir -p $O/r04a_prof_small && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04a_prof_small -o s -- python $R/tools/step_vs_n.py --only 10 > $R/$O/r04a_prof_small/run.log 2>&1)
for c in 63 136; do timeout 600 python tools/cold_path.py $c 2>&1 | grep -v amdgpu.ids > $O/r04a_cold_path_$c.txt; cat $O/r04a_cold_path_$c.txt; done
