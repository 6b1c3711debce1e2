# A/B of the headline step as one fused call against the two calls it replaces, on lattices rattled by SIGMAS (default 0 0.05 0.2);
# WITH_F64=1: the same against the measuring build with double-precision pair tests (make -C mdapy_amd/csrc fcna64)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
clean() { grep -v "^W2\|^E2\|amdgpu.ids\|^\[W\|^\[E"; }
python -m pytest tests/test_gpu_parity.py -x -q -k "fcna or fused or cna" 2>&1 | tail -3
for s in ${SIGMAS:-0.0 0.05 0.2}; do
  echo "=== the library, sigma $s"; python tools/fused_ab.py 136 $s 20 2>&1 | clean
  if [ -n "$WITH_F64" ]; then echo "=== f64 fused (make fcna64), sigma $s"; NB_LIB=mdapy_amd/csrc/libmdapy_amd_fcna64.so python tools/fused_ab.py 136 $s 20 2>&1 | clean; fi
done
