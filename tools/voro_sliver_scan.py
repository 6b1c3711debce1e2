#!/usr/bin/env python
"""Which faces does voro++ have?  The product's Voronoi face routine compiled for the host (tools/voro_sliver_dbg.cpp around
mdapy_amd/csrc/voro_core.hpp) against the reference's voro++ (oracle/_ref) on perfect, barely rattled and rattled lattices and a gas:
atoms whose face count differs under the old rule (area > 1e-14 d^2) and under the width rule, and the range of relative areas and
widths of the small faces.  CPU only; test infrastructure (uses the oracle).
    g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -o /tmp/voro_sliver_dbg.so tools/voro_sliver_dbg.cpp && python tools/voro_sliver_scan.py"""
import sys, ctypes as C
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import numpy as np
from oracle import oracle as O
from mdapy_amd.build_lattice import lattice_positions
D=C.CDLL('/tmp/voro_sliver_dbg.so')
def run(name,pos,L,bd,rc):
    N=len(pos); x,y,z=(np.ascontiguousarray(pos[:,k]) for k in range(3)); bd=np.array(bd,np.int32); org=np.zeros(3)
    v0,n0,r0=np.zeros(N),np.zeros(N,np.int32),np.zeros(N)
    O.get_voronoi_volume_number_radius(x,y,z,np.diag(L),org,bd,v0,n0,r0)
    nA=np.zeros(N,np.int32); nB=np.zeros(N,np.int32); small=np.zeros(3*200000); ns=C.c_int(0)
    rc_=D.dbg_voro(x.ctypes.data_as(C.c_void_p),y.ctypes.data_as(C.c_void_p),z.ctypes.data_as(C.c_void_p),C.c_int64(N),np.ascontiguousarray(L,dtype=float).ctypes.data_as(C.c_void_p),bd.ctypes.data_as(C.c_void_p),org.ctypes.data_as(C.c_void_p),C.c_double(rc),nA.ctypes.data_as(C.c_void_p),nB.ctypes.data_as(C.c_void_p),small.ctypes.data_as(C.c_void_p),C.c_int(200000),C.byref(ns))
    sm=small[:3*ns.value].reshape(-1,3)
    print(f"{name:28s} N {N:5d} rc={rc_}: atoms where the area rule differs from voro++: {(nA!=n0).sum():5d}   width rule: {(nB!=n0).sum():5d}   small faces {len(sm)}"+(f" rel-area {sm[:,1].min():.1e}..{sm[:,1].max():.1e} width {sm[:,2].min():.1e}..{sm[:,2].max():.1e}" if len(sm) else ""), flush=True)
    return sm, nA, nB, n0
rng=np.random.default_rng(3)
for kind,a,n in (("fcc",3.615,5),("bcc",2.87,6),("hcp",2.95,5),("fcc",3.615,5)):
    pos,box=lattice_positions(kind,a,n,n,n); box=np.asarray(box,float); L=np.diag(box) if box.ndim==2 else box
    if box.ndim==2 and np.abs(box-np.diag(np.diag(box))).max()>1e-9: print(kind,"not orthogonal, skipped"); continue
    for sig in (0.0,1e-6,0.01,0.03,0.1):
        for bd in ((1,1,1),(1,1,0)):
            p=pos+rng.normal(0,sig,pos.shape) if sig else pos.copy()
            if not all(bd): p[:,2]=np.clip(p[:,2],1e-3,L[2]-1e-3)
            run(f"{kind} sig {sig} bd {bd}",p,L,bd,2.6*a)
p=rng.random((1500,3))*25.0
run("gas",p,np.array([25.0]*3),(1,1,1),12.0)
