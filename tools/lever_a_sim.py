#!/usr/bin/env python
"""VERDICT round 5, item 6, lever (a) — "chunks formed from centres with equal run-length class ... a wave walks ~76 instead of 108 slots":
what the tile kernel's scan would walk on the headline lattice if every chunk of 64 centres took, per run, 8 candidate slots when no
lane's run holds more than 8 candidates and 12 otherwise — with the centres in cell order (today's chunks) and sorted by their
pattern of long runs.  A simulation on the CPU (numpy), the cell grid as neighbor.cpp:29-62 makes it.  python tools/lever_a_sim.py"""
import numpy as np
a=3.615; rc=0.854*a; n=68
L=a*n
# fcc lattice positions in bench order (x slowest?, basis fastest)
ix,iy,iz=np.meshgrid(np.arange(n),np.arange(n),np.arange(n),indexing='ij')
base=np.stack([ix,iy,iz],-1).reshape(-1,1,3)
basis=np.array([[0,0,0],[.5,.5,0],[.5,0,.5],[0,.5,.5]])
pos=((base+basis[None])*a).reshape(-1,3)
nc=int(np.floor(L/rc))
c=np.minimum((pos/rc).astype(int),nc-1)
pop=np.zeros((nc,nc,nc),int)
np.add.at(pop,(c[:,0],c[:,1],c[:,2]),1)
# run length of cell (x,y,z) = pop[z-1]+pop[z]+pop[z+1] (periodic)
run=pop+np.roll(pop,1,2)+np.roll(pop,-1,2)
print("cells",nc,"mean pop",pop.mean(),"run mean",run.mean(),"P(run>8)",(run>8).mean(),"P(run>12)",(run>12).mean(), "max", run.max())
# per centre cell: 9 runs = run at (x+dx,y+dy,z)
runs9=np.stack([np.roll(np.roll(run,-dx,0),-dy,1) for dx in (-1,0,1) for dy in (-1,0,1)],-1)  # (nc,nc,nc,9)
long9=runs9>8
print("P(all 9 runs <=8 | cell)",(~long9).all(-1).mean(), "mean long runs per cell", long9.sum(-1).mean())
# tiles 4x4x5: centres of a tile (atoms weighted by pop), chunks of 64 in (a) cell order, (b) sorted by mask
def slots_for_tile(t0,t1,t2,sort):
    cells=[(x%nc,y%nc,z%nc) for x in range(t0,t0+4) for y in range(t1,t1+4) for z in range(t2,t2+5)]
    masks=[]
    for (x,y,z) in cells:
        m=long9[x,y,z]
        for _ in range(pop[x,y,z]): masks.append(m)
    masks=np.array(masks)
    if sort:
        key=(masks*(1<<np.arange(9))).sum(1)
        order=np.lexsort((key, masks.sum(1)))
        masks=masks[order]
    tot=0; cen=len(masks)
    for s in range(0,cen,64):
        ch=masks[s:s+64]
        tot+= (np.where(ch.any(0),12,8)).sum()*1.0
    nch=(cen+63)//64
    return tot, nch, cen
rng=np.random.default_rng(0)
for sort in (False,True):
    T=0;C=0
    for _ in range(300):
        t=rng.integers(0,nc,3)
        tot,nch,cen=slots_for_tile(t[0],t[1],t[2],sort)
        T+=tot;C+=nch
    print("sorted by long-run mask" if sort else "cell order", "mean slots per chunk", T/C, "(108 = today's)")
