"""Drop-in for ``mdapy._sfc`` (src/structure_factor.cpp:650-705)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def compute_sfc_direct(x, y, z, box, origin, boundary, structure_factor_py, bins, k_max, k_min, query_x=None, query_y=None,
                       query_z=None, N_total=0, num_t=1):
    """src/structure_factor.cpp:654 — structure_factor_py (bins) host array, NaN for empty bins"""
    if (query_x is None) != (query_y is None) or (query_x is None) != (query_z is None):
        raise ValueError("All query_x, query_y, query_z must be provided or none.")
    if query_x is not None and not N_total:
        raise ValueError("N_total is required when query points are provided.")
    b = np.ascontiguousarray(box, dtype=f64).reshape(9)
    c = Call(x, y, z, query_x, query_y, query_z)
    rc_ = _lib.lib().mdh_sfc_direct(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), int(len(x)), b.ctypes.data,
                                    structure_factor_py.ctypes.data, int(bins), float(k_max), float(k_min), c.inp(query_x, f64),
                                    c.inp(query_y, f64), c.inp(query_z, f64), 0 if query_x is None else int(len(query_x)),
                                    int(N_total), c.space, c.stream)
    c.done(rc_)


def compute_sfc_direct_partial(x, y, z, type_list, Ntype, box, origin, boundary, partial_out, bins, k_max, k_min, num_t=1):
    """src/structure_factor.cpp:682 — partial_out (Ntype, Ntype, bins) host array of Ashcroft-Langreth partials"""
    b = np.ascontiguousarray(box, dtype=f64).reshape(9)
    c = Call(x, y, z, type_list)
    rc_ = _lib.lib().mdh_sfc_direct_partial(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), c.inp(type_list, i32), int(Ntype),
                                            int(len(x)), b.ctypes.data, partial_out.ctypes.data, int(bins), float(k_max),
                                            float(k_min), c.space, c.stream)
    c.done(rc_)
