"""Drop-in for ``mdapy._polycrystal`` (src/polycrystal.cpp:127-143)."""
import ctypes

import numpy as np

from . import _lib
from .devarray import Call

f64 = np.float64


def transform_and_filter(x, y, z, rotation_matrix, center, target_center, coeffs, num_t=1):
    """src/polycrystal.cpp:20 -> (count, 3) f64: the atoms of (x, y, z), rotated about ``center`` and moved to
    ``target_center``, that lie strictly inside every plane of ``coeffs`` (n_faces, 4); input order is kept."""
    rot = np.ascontiguousarray(np.asarray(rotation_matrix, f64).reshape(3, 3))
    c0 = np.ascontiguousarray(np.asarray(center, f64).reshape(3))
    t0 = np.ascontiguousarray(np.asarray(target_center, f64).reshape(3))
    pl = np.ascontiguousarray(np.asarray(coeffs, f64).reshape(-1, 4))
    n = int(len(x))
    _lib.same_rows("transform_and_filter", n, y=y, z=z)
    out = np.empty((n, 3), f64)
    cnt = ctypes.c_int64(0)
    c = Call(x, y, z)
    rc_ = _lib.lib().mdh_transform_and_filter(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, rot.ctypes.data, c0.ctypes.data,
                                              t0.ctypes.data, pl.ctypes.data, int(pl.shape[0]), c.out(out, f64, upload=False),
                                              ctypes.byref(cnt), c.space, c.stream)
    c.done(rc_)
    return out[: int(cnt.value)].copy()
