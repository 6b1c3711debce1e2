"""Drop-in for ``mdapy._wcp`` (src/warren_cowley_parameter.cpp:82-85)."""
import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def get_wcp(verlet_list, neighbor_number, type_list, Ntype, WCP, num_t=1):
    """src/warren_cowley_parameter.cpp:9 — type_list 0-based"""
    c = Call(verlet_list, neighbor_number, type_list, WCP)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_wcp(c.inp(verlet_list, i32), c.inp(neighbor_number, i32), c.inp(type_list, i32), N, M,
                             int(Ntype), c.out(WCP, f64, upload=False), c.space, c.stream)
    c.done(rc_)


def get_wcp_counts(verlet_list, neighbor_number, type_list, Ntype, counts, rows=None):
    """Extension for the multi-GPU path: raw Z_mn | Z_m | atoms-per-type (int64 array of length Nt*Nt+2*Nt; the library writes u64) over the rows with
    ``rows[i] != 0`` (uint8 mask, None = all)."""
    c = Call(verlet_list, neighbor_number, type_list, rows, counts)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_wcp_counts(c.inp(verlet_list, i32), c.inp(neighbor_number, i32), c.inp(type_list, i32),
                                    c.inp(rows, np.uint8), N, M, int(Ntype), c.out(counts, np.int64, upload=False),
                                    c.space, c.stream)
    c.done(rc_)
