# SQ / TA / cache counters of every kernel of the headline step (k_fcna_f32, the cell-grid kernels)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-pmc"
tools/pmc_any.sh step_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $CMD | cut -c1-400
tools/pmc_any.sh step_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $CMD | cut -c1-400
tools/pmc_any.sh step_sq3 "SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" $CMD | cut -c1-400
tools/pmc_any.sh step_ta "TA_TA_BUSY_sum TA_BUSY_avr" $CMD | cut -c1-400
tools/pmc_any.sh step_tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" $CMD | cut -c1-400
tools/pmc_any.sh step_tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" $CMD | cut -c1-400
tools/pmc_any.sh step_mem "FETCH_SIZE WRITE_SIZE" $CMD | cut -c1-400
