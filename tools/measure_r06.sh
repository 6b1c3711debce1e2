#!/bin/bash
# Round-6 measurements, one script with sections: tools/measure_r06.sh <section> [...]  (on the GPU box through gpurun; everything lands
# under gpurun_out/r06_*, tools/make_profiles_r06.py turns it into the tracked files under profiles/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out
clean() { grep -v "^W2\|^E2\|amdgpu.ids\|^\[W\|^\[E"; }
prof() { # prof <tag> <command...> : kernel trace + stats of a command
  local tag=$1; shift
  mkdir -p $O/r06_$tag
  (cd /tmp && PYTHONPATH=$R timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06_$tag -o s -- "$@" > $R/$O/r06_$tag/run.log 2>&1)
  clean < $O/r06_$tag/run.log | tail -2 | cut -c1-300
  python tools/kstats.py $O/r06_$tag 14
}
for section in "$@"; do
echo "=================== section $section"
case $section in
analyses)   # configs 2 and 4 at full size through System under the kernel trace
  prof analyses python $R/tools/profile_analyses.py c3 c5 ;;
analyses_counters)   # SQ / traffic counters of the same command, four separate --pmc passes (kernel trace only: gpurun's rule)
  PROBE="python $R/tools/profile_analyses.py c3 c5"
  tools/pmc_any.sh r06_an_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $PROBE | tail -1
  tools/pmc_any.sh r06_an_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $PROBE | tail -1
  tools/pmc_any.sh r06_an_fetch "FETCH_SIZE" $PROBE | tail -1
  tools/pmc_any.sh r06_an_write "WRITE_SIZE" $PROBE | tail -1 ;;
bench)     # the default bench line, then the same command under the kernel trace
  timeout 1500 python bench.py > $O/r06_bench.json 2> $O/r06_bench.err; tail -c 600 $O/r06_bench.json
  prof bench python $R/bench.py --no-extra --no-pmc --no-cpu-baseline ;;
counters)   # SQ / traffic counters of the headline step's kernels
  PROBE="python $R/tools/order_probe.py lattice 136 5"
  tools/pmc_any.sh r06_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $PROBE | tail -1
  tools/pmc_any.sh r06_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $PROBE | tail -1
  tools/pmc_any.sh r06_fetch "FETCH_SIZE" $PROBE | tail -1
  tools/pmc_any.sh r06_write "WRITE_SIZE" $PROBE | tail -1 ;;
strong)
  prof strong python $R/tools/strong_probe.py 136 20
  clean < $O/r06_strong/run.log | grep slab ;;
knn)
  python tools/knn_split.py 136 2>&1 | clean | tee $O/r06_knn_split.txt
  for r in 1 0; do echo "== MDH_KNN_ROWS=$r"; MDH_KNN_ROWS=$r python tools/knn_rows_ab.py 136 2>&1 | clean; done | tee $O/r06_knn_rows.txt ;;
ptm)
  for lat in fcc bcc hcp gas; do echo "--- $lat"; PTM_LATTICE=$lat PTM_SIGMA=0.05 python tools/ptm_bench.py 136 fcc-hcp-bcc+all 2>&1 | clean | grep -v "^knn"; done | tee $O/r06_ptm_bench.txt ;;
notebook)
  python tools/notebook_probe.py both 3 2>&1 | clean | tee $O/r06_notebook.txt ;;
rc_sweep)
  python tools/rc_sweep.py 2>&1 | clean | tee $O/r06_rc_sweep.txt ;;
tests)
  timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ;;
*) echo "unknown section $section" ;;
esac
done
