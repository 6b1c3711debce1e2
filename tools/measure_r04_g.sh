export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O/r04g_cold
timeout 300 python tools/cold_profile.py 63 2>&1 | grep -v amdgpu.ids > $O/r04g_cold_profile.txt; cat $O/r04g_cold_profile.txt
(cd /tmp && timeout 600 rocprofv3 --hip-trace --output-format csv -d $R/$O/r04g_cold -o c -- python $R/tools/cold_path.py 63 apis > $R/$O/r04g_cold/run.log 2>&1)
