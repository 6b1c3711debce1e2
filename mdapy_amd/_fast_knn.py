"""Drop-in for ``mdapy._fast_knn.knn`` (src/fast_knn.cpp:1024-1031)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def knn(x, y, z, box, origin, boundary, k, indices, distances, num_t=1, key=None):
    """src/fast_knn.cpp:846.  Extension: ``key`` (i64, N; a permutation of 0 .. N-1) orders exact ties in distance by key instead
    of by index (mdh_knn_keyed) — the rows of the system in the key's numbering, neighbour for neighbour."""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, indices, distances, key)
    rc_ = _lib.lib().mdh_knn_keyed(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), int(len(x)), pb, po, pp, int(k),
                                   c.out(indices, i32, upload=False), c.out(distances, f64, upload=False), c.inp(key, np.int64),
                                   c.space, c.stream)
    c.done(rc_)
