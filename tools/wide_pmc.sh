cd $GRAFT_REPO_ROOT
python tools/nb_probe.py 136 50 1.38313 > gpurun_out/wide_nb.txt 2>&1
bash tools/pmc_any.sh wideA "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" python $GRAFT_REPO_ROOT/tools/nb_probe.py 136 50 1.38313
bash tools/pmc_any.sh wideB "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM" python $GRAFT_REPO_ROOT/tools/nb_probe.py 136 50 1.38313
