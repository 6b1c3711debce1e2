#!/bin/bash
# A/B of two builds of the library on one box: tools/ab/lib_old.so against tools/ab/lib_new.so, alternating
L=mdapy_amd/csrc/libmdapy_amd.so
cp $L /tmp/keep.so
for r in 1 2 3; do
  for v in old new; do
    cp tools/ab/lib_$v.so $L
    python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-pmc --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms'].items()}, round(d['roofline']['frac'],4))"
  done
done
cp /tmp/keep.so $L
