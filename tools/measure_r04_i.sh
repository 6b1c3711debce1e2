# unwrapped input through the tile kernel (image codes of up to +-14 box lengths): parity, then the 10 M-atom call
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -m gpu -x -q > $O/r04i_pytest.log 2>&1; tail -3 $O/r04i_pytest.log
for u in 0 1 3 14 20; do echo "unwrap=$u $(NB_UNWRAP=$u python tools/nb_probe.py 136 16 0.854 0.05 10 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-300)"; done | tee $O/r04i_unwrapped.txt
