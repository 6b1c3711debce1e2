#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter list...>   (one rocprofv3 PMC pass over tools/nb_once.py; csv under gpurun_out/pmc_<tag>)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/tools/nb_once.py 136 2 > $R/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag rc=$?"
