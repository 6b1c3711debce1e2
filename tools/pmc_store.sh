#!/bin/bash
PROBE="python $GRAFT_REPO_ROOT/tools/nb_probe.py 136 16 0.854 0.0 5"
tools/pmc_any.sh st1 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES" $PROBE | grep "false, false, false, false, true"
tools/pmc_any.sh st2 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_WRITE_WAVEFRONTS_sum" $PROBE | grep "false, false, false, false, true"
tools/pmc_any.sh st3 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" $PROBE | grep "false, false, false, false, true"
tools/pmc_any.sh st4 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum" $PROBE | grep "false, false, false, false, true"
tools/pmc_any.sh st5 "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $PROBE | grep "false, false, false, false, true"
tools/pmc_any.sh st6 "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_WRITE_sum" $PROBE | grep "false, false, false, false, true"
