#!/bin/bash
# Round-5 measurements, ONE script with sections: tools/measure_r05.sh <section> [...]   (run on the GPU box through gpurun;
# everything lands under gpurun_out/r05_*, the summaries that are judged are copied into profiles/ by hand or by make_profiles)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out
clean() { grep -v "^W2\|^E2\|amdgpu.ids\|^\[W\|^\[E"; }
prof() { # prof <tag> <command...> : kernel trace + stats of a command, top kernels printed
  local tag=$1; shift
  mkdir -p $O/r05_$tag
  (cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r05_$tag -o s -- "$@" > $R/$O/r05_$tag/run.log 2>&1)
  clean < $O/r05_$tag/run.log | tail -2 | cut -c1-600
  python tools/kstats.py $O/r05_$tag 12
}
for section in "$@"; do
echo "=================== section $section"
case $section in
order)     # the headline step on the same atoms in five orders, per-kernel split of each
  for o in lattice blocks shuffled poly poly_shuffled; do echo "--- order $o"; prof order_$o python $R/tools/order_probe.py $o 136 10; done ;;
two_calls)  # the same step as the two calls it was until the labels moved into the tile kernel: kernel split on the lattice and the shuffled frame
  for o in lattice shuffled; do echo "--- order $o (two calls)"; PROBE_TWO_CALLS=1 prof two_$o python $R/tools/order_probe.py $o 136 10; done ;;
fused_ab)   # one call against two on lattices rattled by 0 ... 0.3 A, and against the double-precision form (make fcna64)
  WITH_F64=1 SIGMAS="0.0 0.05 0.1 0.2 0.3" bash tools/ab_fused.sh 2>&1 | tee $O/r05_fused_ab.txt ;;
bench)     # the default bench line, then the same command under the kernel trace
  timeout 1200 python bench.py > $O/r05_bench.json 2> $O/r05_bench.err; tail -c 900 $O/r05_bench.json
  prof bench python $R/bench.py --no-extra --no-pmc --no-cpu-baseline ;;
bench_quick)
  timeout 900 python bench.py --no-cpu-baseline --no-pmc > $O/r05_bench_quick.json 2> $O/r05_bench_quick.err; python - <<P
import json
r = json.loads(open("$O/r05_bench_quick.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["frac"], r["kernels_ms"])
for k, v in r.get("extra", {}).items():
    if k == "reference_notebook_calls":
        for kk, vv in v.items():
            print("  ", kk, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in vv.items()} if isinstance(vv, dict) else vv)
        continue
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "ratio", "ratio_prefetched", "ratio_to_ordered", "ratio_to_builder_order", "kernels_ms", "error")})
P
  ;;
order_sweep)
  python tools/order_sweep.py 136 2>&1 | clean | tee $O/r05_order_sweep.txt ;;
sq_wave)
  prof sq_wave python $R/tools/sq_wave_ab.py 136
  clean < $O/r05_sq_wave/run.log > $O/r05_sq_wave.txt; python tools/kstats.py $O/r05_sq_wave 8 | grep -i "sq_" >> $O/r05_sq_wave.txt; cat $O/r05_sq_wave.txt ;;
lane_tiles)  # the headline build with the tile shape forced (the planner picks 4x4x5): k_neighbor per shape
  for t in "" 4,5 5,3 3,8 4,4 3,7 5,2 6,2; do echo "--- MDH_LANE_TILE=$t"; MDH_LANE_TILE=$t python tools/kbench.py 136 16 0.854 2>&1 | clean | grep -A1 "variant=0" | cut -c1-260; done | tee $O/r05_lane_tiles.txt ;;
strong)   # the 1-of-8 slab step: wall times, then the device timeline of one step
  prof strong python $R/tools/strong_probe.py 136 20
  clean < $O/r05_strong/run.log | grep slab > $O/r05_strong.txt
  python tools/timeline.py $O/r05_strong k_slab_messages 2 >> $O/r05_strong.txt; cat $O/r05_strong.txt ;;
weak_trace)
  prof weak python $R/tools/strong_probe.py 136 10 weak
  python tools/timeline.py $O/r05_weak k_slab_messages 2 | tee $O/r05_weak.txt ;;
halo)   # weak scaling: a 136^3-cell slab of 8 with its halo in loop-back against the undivided 136^3 box
  python tools/halo_cost.py 136 8 2>&1 | clean | tee $O/r05_halo_cost.txt ;;
analyses)   # configs 2 and 4 at full size through System under the kernel trace
  prof analyses python $R/tools/profile_analyses.py c3 c5 ;;
notebook)   # the reference notebook's published workflow call by call, then the kernels behind the list-reuse form
  python tools/notebook_probe.py both 3 2>&1 | clean | tee $O/r05_notebook.txt
  prof notebook python $R/tools/notebook_probe.py reuse 2 | tee -a $O/r05_notebook.txt ;;
consumers)   # the list consumers and secondary analyses at 4 M atoms through System: wall times, then the kernels behind them
  python tools/consumer_times.py 2>&1 | clean | tee $O/r05_consumer_times.txt
  prof consumers python $R/tools/consumer_times.py | tail -24 | tee -a $O/r05_consumer_times.txt ;;
sort)
  python tools/sort_probe.py 2>&1 | clean | tee $O/r05_sort_rows.txt ;;
twin)
  python tools/twin_probe.py 136 2>&1 | clean | tee $O/r05_twin_probe.txt ;;
counters)   # SQ / traffic counters of the headline step's kernels, four separate --pmc passes (kernel trace only: gpurun's rule)
  PROBE="python $R/tools/order_probe.py lattice 136 5"
  tools/pmc_any.sh r05_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $PROBE | tail -1
  tools/pmc_any.sh r05_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $PROBE | tail -1
  tools/pmc_any.sh r05_fetch "FETCH_SIZE" $PROBE | tail -1
  tools/pmc_any.sh r05_write "WRITE_SIZE" $PROBE | tail -1 ;;
tests_new)
  timeout 1500 python -m pytest tests/test_gpu_order.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -15 ;;
tests_dist)
  timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -5 ;;
tests)
  timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ;;
*) echo "unknown section $section" ;;
esac
done
