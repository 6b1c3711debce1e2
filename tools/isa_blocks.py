#!/usr/bin/env python
"""Basic blocks of one kernel in a hipcc -S listing: python tools/isa_blocks.py <file.s> <kernel-substring> [min_instr]
per block: label, instructions, VALU, LDS reads / writes / atomics, global loads / stores, s_waitcnt, and the branches that end it.
Loops show up as a branch back to an earlier label; a small block with one ds_read and one s_waitcnt lgkmcnt(0) that branches
to itself is a dependent-latency loop."""
import re, sys
path, want = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l.split(":")[0] and l.rstrip().endswith(l.split(":")[0].strip()[0:0] + l.split(";")[0].strip()[-1:]) ) if False else None
for i, l in enumerate(lines):
    if l.startswith("_Z") and want in l.split(":")[0]:
        start = i; break
blocks, cur = [], {"label": "entry", "ins": []}
order = {}
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = {"label": m.group(1), "ins": []}; continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["ins"].append(t.split(";")[0].strip())
blocks.append(cur)
for k, b in enumerate(blocks): order[b["label"]] = k
tot = 0
for k, b in enumerate(blocks):
    ins = b["ins"]; tot += len(ins)
    c = lambda pat: sum(1 for x in ins if re.match(pat, x))
    br = [x.split()[-1] for x in ins if x.startswith("s_cbranch") or x.startswith("s_branch")]
    back = [t for t in br if t in order and order[t] <= k]
    if len(ins) < minn and not back:
        continue
    print(f"{k:4d} {b['label']:12s} n={len(ins):4d} valu={c(r'v_'):4d} f64={c(r'v_[a-z0-9_]*f64'):4d} dsr={c(r'ds_read'):3d} dsw={c(r'ds_write'):3d} dsa={c(r'ds_(or|add|and|max|min|cmpst|wrxchg)'):2d} "
          f"gld={c(r'(global|flat|buffer)_load'):3d} gst={c(r'(global|flat|buffer)_store'):3d} wait={c(r's_waitcnt'):3d} br={','.join(br)}{'  <== LOOP to ' + ','.join(back) if back else ''}")
print("total instructions", tot)
