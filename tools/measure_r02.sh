export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -c 1500 gpurun_out/r02_bench.json
mkdir -p gpurun_out/r02_prof_bench && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_bench -o b -- python $R/bench.py --no-extra --no-pmc > $R/gpurun_out/r02_prof_bench/run.log 2>&1)
head -8 gpurun_out/r02_prof_bench/b_kernel_stats.csv | cut -c1-60,300-420
PROBE="python $R/tools/nb_probe.py 136 16 0.854 0 2"
tools/pmc_any.sh nb_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $PROBE | tail -3
tools/pmc_any.sh nb_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $PROBE | tail -3
tools/pmc_any.sh nb_fetch "FETCH_SIZE" $PROBE | tail -3
tools/pmc_any.sh nb_write "WRITE_SIZE" $PROBE | tail -3
mkdir -p gpurun_out/r02_prof_an && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_an -o an -- python $R/tools/profile_analyses.py c3 c5 > $R/gpurun_out/r02_prof_an/run.log 2>&1)
grep -v "^W2\|^E2" gpurun_out/r02_prof_an/run.log | tail -30
tools/pmc_any.sh an_fetch "FETCH_SIZE" python $R/tools/profile_analyses.py c3 c5 | tail -2
tools/pmc_any.sh an_write "WRITE_SIZE" python $R/tools/profile_analyses.py c3 c5 | tail -2
mkdir -p gpurun_out/r02_prof_reader && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_reader -o rd -- python $R/tools/reader_probe.py > $R/gpurun_out/r02_prof_reader/run.log 2>&1)
grep "read " gpurun_out/r02_prof_reader/run.log
tools/pmc_any.sh an_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" python $R/tools/profile_analyses.py c3 c5 | tail -1
tools/pmc_any.sh an_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" python $R/tools/profile_analyses.py c3 c5 | tail -1
python tools/nb_probe.py 136 16 0.854 0 5 0.1 | tail -1 > gpurun_out/r02_tri_lane.txt; NB_VARIANT=1 python tools/nb_probe.py 136 16 0.854 0 3 0.1 | tail -1 >> gpurun_out/r02_tri_lane.txt; cat gpurun_out/r02_tri_lane.txt
python tools/ptm_bench.py 136 fcc-hcp-bcc+all 2>&1 | tail -4 > gpurun_out/r02_ptm_bench.txt; cat gpurun_out/r02_ptm_bench.txt
