#!/bin/bash
# SQ counter passes over the lane-kernel probe -> gpurun_out/pmc_<tag>*.json
tag=${1:-lane}
PROBE="python $GRAFT_REPO_ROOT/tools/nb_probe.py 136 16 0.854 0.0 5"
tools/pmc_any.sh ${tag}_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" $PROBE | grep k_neighbor_lane
tools/pmc_any.sh ${tag}_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $PROBE | grep k_neighbor_lane
tools/pmc_any.sh ${tag}_sq3 "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" $PROBE | grep k_neighbor_lane
