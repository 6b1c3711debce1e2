# round-4 measurement pass (one gpurun call): bash tools/measure_r04.sh ; then python tools/make_profiles_r04.py in the build container
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out
timeout 900 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err; tail -c 600 $O/r04_bench.json
mkdir -p $O/r04_prof_bench && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04_prof_bench -o b -- python $R/bench.py --no-extra --no-pmc --no-cpu-baseline > $R/$O/r04_prof_bench/run.log 2>&1)
PROBE="python $R/tools/nb_probe.py 136 16 0.854 0 2"
tools/pmc_any.sh nb_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $PROBE | tail -2
tools/pmc_any.sh nb_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $PROBE | tail -2
tools/pmc_any.sh nb_fetch "FETCH_SIZE" $PROBE | tail -2
tools/pmc_any.sh nb_write "WRITE_SIZE" $PROBE | tail -2
# phase stamps of the tile kernel: headline (four and three workgroups per CU), the rc 5 / 50-slot call
{
echo "== headline: 136^3 fcc Cu, rc 0.854 a, M 16 (four workgroups per CU)"; NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 136 16 0.854 0.0 5 2>&1 | grep -v amdgpu.ids
echo "== the same with the LDS cut for three workgroups per CU"; MDH_LANE_WGS=3 NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 136 16 0.854 0.0 5 2>&1 | grep -v amdgpu.ids
echo "== rc 5.0, M 50 (wide instance), 10 M atoms"; NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 136 50 1.38313 0.0 3 2>&1 | grep -v amdgpu.ids
} > $O/r04_lane_phase_stamps.txt 2>&1
# the reference's own benchmark call, build_neighbor(5.0, max_neigh=50): 3.4 M and 10 M atoms; thread-per-atom kernel beside it
{
for c in 95 136; do
  echo "== $c^3 cells, tile kernel"; python tools/nb_probe.py $c 50 1.38313 0.0 5 2>&1 | grep -v amdgpu.ids
  echo "== $c^3 cells, thread-per-atom kernel (round 2's path for this call)"; NB_VARIANT=1 python tools/nb_probe.py $c 50 1.38313 0.0 3 2>&1 | grep -v amdgpu.ids
done
} > $O/r04_rc5_m50.txt 2>&1
tail -4 $O/r04_rc5_m50.txt
python tools/consumer_times.py 2>&1 | grep -v amdgpu.ids > $O/r04_consumer_times.txt; tail -3 $O/r04_consumer_times.txt
python tools/halo_cost.py 136 8 2>&1 | grep -v amdgpu.ids > $O/r04_halo_cost.txt; head -3 $O/r04_halo_cost.txt
python tools/host_path.py 2>&1 | grep -v amdgpu.ids > $O/r04_host_path.txt; tail -1 $O/r04_host_path.txt
python tools/nb_probe.py 136 16 0.854 0 5 0.1 | tail -1 > $O/r04_tri_lane.txt; NB_VARIANT=1 python tools/nb_probe.py 136 16 0.854 0 3 0.1 | tail -1 >> $O/r04_tri_lane.txt; cat $O/r04_tri_lane.txt
mkdir -p $O/r04_prof_an && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04_prof_an -o an -- python $R/tools/profile_analyses.py c3 c5 > $R/$O/r04_prof_an/run.log 2>&1)
grep -v "^W2\|^E2" $O/r04_prof_an/run.log | tail -24
tools/pmc_any.sh an_fetch "FETCH_SIZE" python $R/tools/profile_analyses.py c3 c5 | tail -1
tools/pmc_any.sh an_write "WRITE_SIZE" python $R/tools/profile_analyses.py c3 c5 | tail -1
tools/pmc_any.sh an_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" python $R/tools/profile_analyses.py c3 c5 | tail -1
tools/pmc_any.sh an_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" python $R/tools/profile_analyses.py c3 c5 | tail -1
python tools/ptm_bench.py 136 fcc-hcp-bcc+all 2>&1 | tail -4 > $O/r04_ptm_bench.txt; cat $O/r04_ptm_bench.txt
# ---- round 4: fixed cost per step, cold path, the fast-path holes
timeout 600 python tools/step_vs_n.py --json $O/r04_step_vs_n.json 2>&1 | grep -v amdgpu.ids > $O/r04_step_vs_n.txt; tail -8 $O/r04_step_vs_n.txt | cut -c1-300
mkdir -p $O/r04_prof_small && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/r04_prof_small -o s -- python $R/tools/step_vs_n.py --only 10 > $R/$O/r04_prof_small/run.log 2>&1)
for k in 4 11 16; do ./tools/ubench/launch_gap $k 2000 64; done > $O/r04_launch_gap.txt 2>&1
./tools/ubench/atomic_scope > $O/r04_atomic_scope.txt 2>&1
for c in 63 136; do timeout 600 python tools/cold_path.py $c 2>&1 | grep -v amdgpu.ids > $O/r04_cold_path_$c.txt; done
MDAPY_HIP_WARM=0 timeout 600 python tools/cold_path.py 136 apis 2>&1 | grep -v amdgpu.ids > $O/r04_cold_path_136_nowarm.txt
MDAPY_HIP_WARM=0 timeout 600 python tools/cold_path.py 63 apis 2>&1 | grep -v amdgpu.ids > $O/r04_cold_path_63_nowarm.txt
{
for u in 0 1 3 13 20; do echo "unwrapped by up to $u box lengths: $(NB_UNWRAP=$u python tools/nb_probe.py 136 16 0.854 0.05 10 2>&1 | grep -v amdgpu.ids | tail -1)"; done
for pb in 111 101 010; do for var in 0 1; do echo "sheared box (10 %), pbc $pb, $( [ $var = 0 ] && echo 'tile kernel' || echo 'thread-per-atom kernel' ): $(NB_PBC=$pb NB_VARIANT=$var python tools/nb_probe.py 136 16 0.854 0.05 5 0.1 2>&1 | grep -v amdgpu.ids | tail -1)"; done; done
} > $O/r04_fast_path_holes.txt 2>&1
cat $O/r04_fast_path_holes.txt | cut -c1-200
# ---- sweeps over input kinds (where a frame falls off a fast path), the host side of a small step, the first partial RDF, CNA on hot lattices
python tools/unwrapped_sweep.py 100 2>&1 | grep -v amdgpu.ids > $O/r04_unwrapped_sweep.txt
python tools/triclinic_sweep.py 100 2>&1 | grep -v amdgpu.ids > $O/r04_triclinic_sweep.txt
python tools/triclinic_sweep.py 100 --open 2>&1 | grep -v amdgpu.ids > $O/r04_open_box_sweep.txt
for c in 10 20 40; do python tools/host_enqueue.py $c 2>&1 | grep -v amdgpu.ids; done > $O/r04_host_enqueue.txt
python tools/cold_rdf.py 2>&1 | grep -v amdgpu.ids | grep -v "^ \+[0-9]\+ " > $O/r04_cold_rdf.txt
# (round 4 ran a one-off script here: fixed-cutoff CNA at sigma 0 / 0.2 / 0.5 under the kernel trace -> r04_fcna_hot.txt; round 5: tools/measure_r05.sh has sections instead)
tail -4 $O/r04_triclinic_sweep.txt | cut -c1-200
python tools/triclinic_sweep.py 100 --disorder 2>&1 | grep -v amdgpu.ids > $O/r04_disorder_sweep.txt
