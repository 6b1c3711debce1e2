"""Drop-in for ``mdapy._cna`` (src/cna.cpp:508-513): acna, fcna, ids."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def fcna(x, y, z, box, origin, boundary, verlet_list, neighbor_number, pattern, rc, num_t=1):
    """src/cna.cpp:429 — pattern must be pre-zeroed"""
    _lib.same_rows("fcna", len(x), y=y, z=z, verlet_list=verlet_list, neighbor_number=neighbor_number, pattern=pattern)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, neighbor_number, pattern)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_fcna(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.inp(verlet_list, i32), M,
                              c.inp(neighbor_number, i32), c.out(pattern, i32), float(rc), c.space, c.stream)
    c.done(rc_)


def acna(x, y, z, box, origin, boundary, verlet_list, pattern, num_t=1):
    """src/cna.cpp:289 — rows sorted by distance, >= 14 columns"""
    _lib.same_rows("acna", len(x), y=y, z=z, verlet_list=verlet_list, pattern=pattern)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, pattern)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_acna(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.inp(verlet_list, i32), M,
                              c.out(pattern, i32), c.space, c.stream)
    c.done(rc_)


def ids(x, y, z, box, origin, boundary, verlet_list, new_verlet_list, pattern, num_t=1):
    """src/cna.cpp:163"""
    _lib.same_rows("ids", len(x), y=y, z=z, verlet_list=verlet_list, new_verlet_list=new_verlet_list, pattern=pattern)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, new_verlet_list, pattern)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_ids(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.inp(verlet_list, i32), M,
                             c.out(new_verlet_list, i32), c.out(pattern, i32), c.space, c.stream)
    c.done(rc_)
