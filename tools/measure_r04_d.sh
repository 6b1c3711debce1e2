# the hits' coordinates gathered through L2 instead of read from LDS (measuring build, make gather): A/B on one box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for r in 1 2 3; do for v in "" mdapy_amd/csrc/libmdapy_amd_gather.so; do
  echo "lib=${v:-product} $(NB_LIB=$v python tools/nb_probe.py 136 16 0.854 0 20 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-300)"
done; done | tee $O/r04d_gather_ab.txt
NB_LIB=mdapy_amd/csrc/libmdapy_amd_gather.so timeout 600 python - <<'P' 2>&1 | tail -3 | tee -a $O/r04d_gather_ab.txt
import os, sys
sys.path.insert(0, os.getcwd())
from mdapy_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['NB_LIB'])
import pytest
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-k", "neighbor"]))
P
