"""Drop-in for ``mdapy._aja`` (src/ackland_jones_analysis.cpp:174-177)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def compute_aja(x, y, z, box, origin, boundary, verlet_list, distance_list, aja, num_t=1):
    """src/ackland_jones_analysis.cpp:9 — rows hold >= 14 neighbours sorted by distance"""
    _lib.same_rows("compute_aja", len(x), y=y, z=z, verlet_list=verlet_list, distance_list=distance_list, aja=aja)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, distance_list, aja)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_aja(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.inp(verlet_list, i32),
                             c.inp(distance_list, f64), M, c.out(aja, i32, upload=False), c.space, c.stream)
    c.done(rc_)
