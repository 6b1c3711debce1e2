"""Drop-in for ``mdapy._sbo`` (src/steinhardt_bond_orientation.cpp:786-790)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def get_sq(x, y, z, box, origin, boundary, verlet_list, distance_list, neighbor_number, weight, llist, nnn, lmax, wl,
           wlhat, average, use_voronoi, rc, use_weight, qlm_r, qlm_i, qnarray, num_t=1):
    """src/steinhardt_bond_orientation.cpp:677"""
    _lib.same_rows("get_sq", len(x), y=y, z=z, verlet_list=verlet_list, distance_list=distance_list, neighbor_number=neighbor_number)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    ll = np.ascontiguousarray(np.asarray(llist), dtype=i32)
    w = weight if use_weight else None
    c = Call(x, y, z, verlet_list, distance_list, neighbor_number, w, qlm_r, qlm_i, qnarray)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_get_sq(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.inp(verlet_list, i32),
                                c.inp(distance_list, f64), M, c.inp(neighbor_number, i32), c.inp(w, f64),
                                ll.ctypes.data, int(ll.shape[0]), int(nnn), int(lmax), int(bool(wl)),
                                int(bool(wlhat)), int(bool(average)), int(bool(use_voronoi)), float(rc),
                                int(bool(use_weight)), c.out(qlm_r, f64), c.out(qlm_i, f64), c.out(qnarray, f64),
                                c.space, c.stream)
    c.done(rc_)


def identifySolidLiquid(Q6index, Q6, verlet_list, distance_list, neighbor_number, qlm_r, qlm_i, threshold, n_bond,
                        solidliquid, nbond, use_voronoi, nnn, rc, num_t=1):
    """src/steinhardt_bond_orientation.cpp:578"""
    c = Call(Q6, verlet_list, distance_list, neighbor_number, qlm_r, qlm_i, solidliquid, nbond)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_identify_solid_liquid(int(Q6index), c.inp(Q6, f64), c.inp(verlet_list, i32),
                                               c.inp(distance_list, f64), c.inp(neighbor_number, i32), N, M,
                                               c.inp(qlm_r, f64), c.inp(qlm_i, f64), int(qlm_r.shape[1]),
                                               int(qlm_r.shape[2]), float(threshold), int(n_bond),
                                               c.out(solidliquid, i32), c.out(nbond, i32, upload=False),
                                               int(bool(use_voronoi)), int(nnn), float(rc), c.space, c.stream)
    c.done(rc_)
