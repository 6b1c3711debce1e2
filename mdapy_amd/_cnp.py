"""Drop-in for ``mdapy._cnp`` (src/common_neighbor_parameter.cpp:139-142)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def compute_cnp(x, y, z, box, origin, boundary, verlet_list, distance_list, neighbor_number, cnp, rc, num_t=1):
    """src/common_neighbor_parameter.cpp:10"""
    _lib.same_rows("compute_cnp", len(x), y=y, z=z, verlet_list=verlet_list, distance_list=distance_list, neighbor_number=neighbor_number, cnp=cnp)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, distance_list, neighbor_number, cnp)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_cnp(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.inp(verlet_list, i32),
                             c.inp(distance_list, f64), c.inp(neighbor_number, i32), M, c.out(cnp, f64, upload=False),
                             float(rc), c.space, c.stream)
    c.done(rc_)
