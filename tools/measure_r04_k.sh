export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "neighbor or unwrapped or variants" 2>&1 | tail -5
# 10 M atoms in a sheared box, periodic and open along b: tile kernel against the thread-per-atom kernel
for pb in "" "101"; do for var in 0 1; do echo "tri pbc=${pb:-111} variant=$var $(NB_PBC=$pb NB_VARIANT=$var python tools/nb_probe.py 136 16 0.854 0.05 5 0.1 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-300)"; done; done | tee $O/r04k_tri_open.txt
