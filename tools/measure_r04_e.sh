# where the first calls of a process spend their time: HIP API trace of tools/cold_path.py (apis only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O/r04e_cold
(cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $R/$O/r04e_cold -o c -- python $R/tools/cold_path.py 136 apis > $R/$O/r04e_cold/run.log 2>&1)
grep -v amdgpu.ids $O/r04e_cold/run.log | tail -12
ls -la $O/r04e_cold | head; find $O/r04e_cold -name "*hip_api_stats*" | head -2 | xargs -I{} head -25 {}
rm -f $O/r04e_cold/*kernel_trace.csv   # (keep the api trace)
