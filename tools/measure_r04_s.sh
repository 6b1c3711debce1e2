export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "cna or fcna or config or neighbor or host" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-pmc > $O/r04s_bench.json 2>$O/r04s_bench.err; python - <<P
import json
d=json.loads(open("$O/r04s_bench.json").read().strip().splitlines()[-1])
print(d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
for k,v in d['extra'].items(): print(k, v.get('ms_per_step'), v.get('kernels_ms'), v.get('todo_fraction'), v.get('ratio'))
P
