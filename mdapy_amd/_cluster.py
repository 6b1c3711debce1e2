"""Drop-in for ``mdapy._cluster`` (src/cluster.cpp:150-155)."""
import ctypes

import numpy as np

from . import _lib
from .devarray import Call

f64, i32 = np.float64, np.int32


def _run(verlet_list, distance_list, neighbor_number, rc, by_bond, particleClusters):
    c = Call(verlet_list, distance_list, neighbor_number, particleClusters)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    n = ctypes.c_int(0)
    rc_ = _lib.lib().mdh_cluster(c.inp(verlet_list, i32), c.inp(distance_list, f64), c.inp(neighbor_number, i32), N, M,
                                 float(rc), int(by_bond), c.out(particleClusters, i32, upload=False), ctypes.byref(n),
                                 c.space, c.stream)
    c.done(rc_)
    return int(n.value)


def get_cluster(verlet_list, distance_list, neighbor_number, rc, particleClusters):
    """src/cluster.cpp:9 — returns the number of clusters; ids start at 1"""
    return _run(verlet_list, distance_list, neighbor_number, rc, 0, particleClusters)


def get_cluster_by_bond(verlet_list, neighbor_number, particleClusters):
    """src/cluster.cpp:58"""
    return _run(verlet_list, None, neighbor_number, 0.0, 1, particleClusters)


def filter_by_type(verlet_list, distance_list, neighbor_number, type_list, type1, type2, r, num_t=1):
    """src/cluster.cpp:108 — verlet_list is modified in place"""
    t1 = np.ascontiguousarray(type1, dtype=i32)
    t2 = np.ascontiguousarray(type2, dtype=i32)
    rr = np.ascontiguousarray(r, dtype=f64)
    c = Call(verlet_list, distance_list, neighbor_number, type_list)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    _lib.same_rows("filter_by_type", N, distance_list=distance_list, neighbor_number=neighbor_number, type_list=type_list)
    rc_ = _lib.lib().mdh_filter_by_type(c.out(verlet_list, i32), c.inp(distance_list, f64), c.inp(neighbor_number, i32),
                                        c.inp(type_list, i32), N, M, t1.ctypes.data, t2.ctypes.data, rr.ctypes.data,
                                        len(t1), c.space, c.stream)
    c.done(rc_)
