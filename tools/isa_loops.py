import re,sys
path=sys.argv[1]
lines=open(path).read().split("\n")
name=None; blocks=[]; cur=None
out=[]
def flush(name, blocks):
    order={b["label"]:k for k,b in enumerate(blocks)}
    for k,b in enumerate(blocks):
        ins=b["ins"]
        c=lambda pat: sum(1 for x in ins if re.match(pat,x))
        br=[x.split()[-1] for x in ins if x.startswith("s_cbranch") or x.startswith("s_branch")]
        back=[t for t in br if t in order and order[t]<=k]
        loads=c(r'ds_read')+c(r'(global|flat|buffer|scratch)_load')
        if back and len(ins)<=45 and loads>=1 and c(r's_waitcnt')>=1:
            out.append((name[:70], b["label"], len(ins), c(r'ds_read'), c(r'(global|flat|buffer|scratch)_load'), c(r's_waitcnt')))
for l in lines:
    if l.startswith("_Z") and ":" in l and "@" in l:
        name=l.split(":")[0]; blocks=[]; cur={"label":"entry","ins":[]}; continue
    if name is None: continue
    if l.startswith(".Lfunc_end"):
        blocks.append(cur); flush(name, blocks); name=None; continue
    m=re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur={"label":m.group(1),"ins":[]}; continue
    t=l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur["ins"].append(t.split(";")[0].strip())
import collections
agg=collections.Counter()
for o in out: agg[o[0]]+=1
for o in out: print(o)
