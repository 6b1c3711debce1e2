# cold path with the code objects loaded at the library's first use (mdh_warm), against MDAPY_HIP_WARM=0
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for c in 63 136; do
  timeout 600 python tools/cold_path.py $c 2>&1 | grep -v amdgpu.ids > $O/r04f_cold_path_$c.txt; cat $O/r04f_cold_path_$c.txt
done
MDAPY_HIP_WARM=0 timeout 600 python tools/cold_path.py 136 apis 2>&1 | grep -v amdgpu.ids > $O/r04f_cold_path_136_nowarm.txt; cat $O/r04f_cold_path_136_nowarm.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04f_pytest.log 2>&1; tail -3 $O/r04f_pytest.log
