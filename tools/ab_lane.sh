#!/bin/bash
# A/B of the lane neighbour kernel on one box: parity subset, then probe timings under the environment switches.
# usage (on the GPU box): bash tools/ab_lane.sh [out]
out=${1:-gpurun_out/ab_lane.txt}
mkdir -p $(dirname $out)
{
python -m pytest tests/test_gpu_parity.py -x -q -k "neighbor or fused or config1" 2>&1 | tail -5
for w in 4 3; do
  echo "== MDH_LANE_WGS=$w"; MDH_LANE_WGS=$w python tools/nb_probe.py 136 16 0.854 0.0 10 2>&1 | grep -v amdgpu.ids
done
echo "== sigma 0.05"; python tools/nb_probe.py 136 16 0.854 0.05 10 2>&1 | grep -v amdgpu.ids
echo "== stamps"; NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 136 16 0.854 0.0 5 2>&1 | grep -v amdgpu.ids
echo "== stamps wgs3"; MDH_LANE_WGS=3 NB_LIB=mdapy_amd/csrc/libmdapy_amd_stamps.so python tools/nb_probe.py 136 16 0.854 0.0 5 2>&1 | grep -v amdgpu.ids
} > $out 2>&1
tail -40 $out
