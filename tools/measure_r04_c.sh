# NW = 8 big-tile instance of the lane kernel: parity, then A/B against NW = 4 through MDH_LANE_NW on one box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "neighbor or build or overflow or mop or stale or hint or exact or dense or fcna or cna" > $O/r04c_pytest.log 2>&1; tail -3 $O/r04c_pytest.log
for r in 1 2 3; do for nw in 4 8; do
  echo "NW=$nw $(MDH_LANE_NW=$nw python tools/nb_probe.py 136 16 0.854 0 20 2>/dev/null | tr '\n' ' ' | cut -c1-400)"
done; done | tee $O/r04c_ab.txt
for nw in 4 8; do echo "NW=$nw sigma .05 $(MDH_LANE_NW=$nw python tools/nb_probe.py 136 16 0.854 0.05 20 2>/dev/null | tr '\n' ' ' | cut -c1-400)"; done | tee -a $O/r04c_ab.txt
for nw in 4 8; do echo "NW=$nw 63 cells $(MDH_LANE_NW=$nw python tools/nb_probe.py 63 16 0.854 0 50 2>/dev/null | tr '\n' ' ' | cut -c1-400)"; done | tee -a $O/r04c_ab.txt
for nw in 4 8; do echo "NW=$nw rc 1.0a M24 $(MDH_LANE_NW=$nw python tools/nb_probe.py 136 24 1.0 0.05 10 2>/dev/null | tr '\n' ' ' | cut -c1-400)"; done | tee -a $O/r04c_ab.txt
