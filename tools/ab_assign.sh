export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "neighbor" 2>&1 | tail -2
for K in 1 2 4; do for o in lattice poly shuffled; do
  echo "--- K=$K order=$o"
  mkdir -p $O/ab_assign_${K}_$o
  (cd /tmp && MDH_ASSIGN_K=$K PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ab_assign_${K}_$o -o s -- python $R/tools/order_probe.py $o 136 10 > $R/$O/ab_assign_${K}_$o/run.log 2>&1)
  python tools/kstats.py $O/ab_assign_${K}_$o 12 | grep -E "k_assign|k_gather|k_scatter|k_scan|k_sort_cells"
done; done
