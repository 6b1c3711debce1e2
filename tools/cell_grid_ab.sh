#!/bin/bash
# neighbor builds without k_gather — the kernels read the atoms through the cell-sorted id list
# (MDH_INDIRECT=1, the product) against the 32-byte records of k_gather (MDH_INDIRECT=0).  cell_grid / k_neighbor: HIP-event
# ranges inside the library, ms per `reps` builds; the fingerprint of the rows must not depend on the switch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "MDH_INDIRECT=$1 [$2]: $(MDH_INDIRECT=$1 NB_SUM=1 ${4:-} python tools/nb_probe.py $3 2>&1 | grep -E 'cell_grid|k_neighbor|fingerprint' | tr '\n' ' ')"; }
for rep in 1 2 3; do for i in 0 1; do run $i "headline lattice, 10 M atoms, M=16, 20 builds" "136 16 0.854 0.0 20"; done; done
for i in 0 1; do
  run $i "rattled sigma=0.05 (atoms outside the box: image codes)" "136 16 0.854 0.05 20"
  run $i "build_neighbor(5.0, 50), 5 builds" "136 50 1.38313 0.0 5"
  run $i "sheared box" "100 16 0.854 0.02 10 0.2"
  run $i "unwrapped trajectory (atoms up to 2 box lengths away)" "100 16 0.854 0.02 10" "env NB_UNWRAP=2"
  run $i "open along b" "100 16 0.854 0.02 10" "env NB_PBC=101"
done
for i in 0 1; do echo "MDH_INDIRECT=$i bench: $(MDH_INDIRECT=$i python bench.py --steps 200 --warmup 30 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); e=d["extra"]; print(d["ms_per_step"], d["roofline"]["frac"], e.get("kernels_ms"), "polycrystal", e.get("polycrystal",{}).get("ms_per_step"), e.get("polycrystal",{}).get("kernels_ms"))')"; done
