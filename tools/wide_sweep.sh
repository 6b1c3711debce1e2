#!/bin/bash
# the wide instance (build_neighbor(5.0, 50), 10 M atoms): tile shapes and workgroups per CU, forced through the planner's A/B switches
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 1 2 3; do
  for t in "" 2,2 2,3 2,4 2,6 3,1 3,2 3,3 3,4 4,1 4,2; do
    echo "wgs=$w tile=$t: $(MDH_LANE_WGS=$w MDH_LANE_TILE=$t python tools/nb_probe.py 136 50 1.38313 2>&1 | grep -E 'plan|k_neighbor' | tr '\n' ' ')"
  done
done
