"""Drop-in for ``mdapy._csp`` (src/centro_symmetry_parameter.cpp:96-99)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def get_csp(x, y, z, box, origin, boundary, verlet_list, N, csp, num_t=1):
    """src/centro_symmetry_parameter.cpp:12"""
    _lib.same_rows("get_csp", len(x), y=y, z=z, verlet_list=verlet_list, csp=csp)
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, csp)
    n, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_csp(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, c.inp(verlet_list, i32), M,
                             int(N), c.out(csp, f64, upload=False), c.space, c.stream)
    c.done(rc_)
