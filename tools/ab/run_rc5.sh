#!/bin/bash
# A/B of two builds on the reference's own benchmark call, build_neighbor(5.0, max_neigh=50), 10 M atoms
L=mdapy_amd/csrc/libmdapy_amd.so
cp $L /tmp/keep.so
for r in 1 2 3; do
  for v in old new; do
    cp tools/ab/lib_$v.so $L
    echo -n "$v "; python tools/nb_probe.py 136 50 1.38313 0.0 5 2>&1 | grep "^k_neighbor"
  done
done
cp /tmp/keep.so $L
