"""Drop-in for ``mdapy._fast_knn.knn`` (src/fast_knn.cpp:1024-1031)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32
keeps_candidates = True  # knn(..., candidates=dict) (knn.py asks before passing it: the test backends have no such argument)


def knn(x, y, z, box, origin, boundary, k, indices, distances, num_t=1, key=None, candidates=None):
    """src/fast_knn.cpp:846.  Extensions: ``key`` (i64, N; a permutation of 0 .. N-1) orders exact ties in distance by key instead
    of by index (mdh_knn_keyed) — the rows of the system in the key's numbering, neighbour for neighbour.  ``candidates``: a dict the
    caller keeps with THESE positions in THIS box (frame.py: a Frame's columns never change) — the candidate rows of the search's
    cutoff build are left in it and the next search of the same positions, for whatever k, skips that build
    (mdh_knn_keyed_rows).  Same results with or without."""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, indices, distances, key)
    L = _lib.lib()
    n = int(len(x))
    if candidates is None or c.space != _lib.DEVICE:
        rc_ = L.mdh_knn_keyed(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, int(k),
                              c.out(indices, i32, upload=False), c.out(distances, f64, upload=False), c.inp(key, np.int64),
                              c.space, c.stream)
        c.done(rc_)
        return
    import ctypes

    from .devarray import HArray

    sig = (pb, po, pp, n)  # (host_box memoises equal boxes: equal pointers mean equal bytes)
    have = candidates.get("rows")
    radius = ctypes.c_double(0.0)
    if have is not None and candidates.get("sig") == sig:
        rows, counts, radius.value = have
    else:
        width = int(L.mdh_knn_rows_width(int(k)))
        if width <= 0:
            rows = counts = None
        else:
            rows, counts = HArray.empty((n, width), i32), HArray.empty((n,), i32)
    if rows is None:
        rc_ = L.mdh_knn_keyed(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, int(k),
                              c.out(indices, i32, upload=False), c.out(distances, f64, upload=False), c.inp(key, np.int64),
                              c.space, c.stream)
        c.done(rc_)
        return
    rc_ = L.mdh_knn_keyed_rows(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, int(k),
                               c.out(indices, i32, upload=False), c.out(distances, f64, upload=False), c.inp(key, np.int64),
                               rows.dev().data_ptr(), counts.dev().data_ptr(), int(rows.shape[1]), ctypes.addressof(radius),
                               c.space, c.stream)
    c.done(rc_)
    candidates.clear()
    if radius.value > 0.0:
        candidates.update(sig=sig, rows=(rows, counts, float(radius.value)), keep=keep)
