"""Atom order (csrc/order.hip; include/mdapy_amd.h "atom order"): how far the order atoms were handed in is from a spatial
one, a cell sort, and the translation of per-atom columns and list rows between the two index spaces.  No reference
counterpart — the reference's cell build and consumers are indifferent to atom order (src/neighbor.cpp:64-100); these
kernels' gathers are not.  Used by ``System`` for its cell-sorted twin."""
import ctypes

import numpy as np

from . import _lib
from .devarray import Call, HArray, empty

f64, i32 = np.float64, np.int32


def order_statistic(x, y, z, box, origin, boundary):
    """fraction of consecutive atoms (i, i+1) that are not in the same or in touching ~64-atom bins"""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z)
    out = ctypes.c_double(0.0)
    c.done(_lib.lib().mdh_order_statistic(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), int(len(x)), pb, po, pp, ctypes.addressof(out),
                                          c.space, c.stream))
    return float(out.value)


def spatial_sort(x, y, z, box, origin, boundary):
    """-> (xs, ys, zs, perm, n_sorted): the positions in cell order and the atom each sorted slot holds; n_sorted < N when
    absent atoms (x = NaN) were handed in (perm is then no permutation)"""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    N = int(len(x))
    on_dev = not all(isinstance(a, np.ndarray) for a in (x, y, z))
    mk = (lambda dt: HArray.empty((N,), dt)) if on_dev else (lambda dt: np.empty(N, dt))
    xs, ys, zs, perm = mk(f64), mk(f64), mk(f64), mk(i32)
    n = ctypes.c_int64(0)
    c = Call(x, y, z, xs, ys, zs, perm)
    c.done(_lib.lib().mdh_spatial_sort(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, c.out(xs, f64, upload=False),
                                       c.out(ys, f64, upload=False), c.out(zs, f64, upload=False), c.out(perm, i32, upload=False),
                                       ctypes.addressof(n), c.space, c.stream))
    return xs, ys, zs, perm, int(n.value)


def permute(values, perm, scatter=False):
    """out[p] = values[perm[p]] (gather) or out[perm[p]] = values[p] (scatter), 4- or 8-byte numeric columns"""
    dt = np.dtype(values.dtype)
    if dt.itemsize not in (4, 8) or dt.kind not in "iuf":
        raise TypeError(f"permute: 4- or 8-byte numeric columns, got {dt}")
    N = int(len(values))
    on_dev = not (isinstance(values, np.ndarray) and isinstance(perm, np.ndarray))
    out = HArray.empty((N,), dt) if on_dev else np.empty(N, dt)
    c = Call(values, perm, out)
    c.done(_lib.lib().mdh_permute(c.inp(values, dt), c.inp(perm, i32), N, dt.itemsize, int(bool(scatter)), c.out(out, dt, upload=False),
                                  c.space, c.stream))
    return out


def gather_positions(x, y, z, perm):
    """(x[perm], y[perm], z[perm]) in one pass (mdh_gather_positions)"""
    N = int(len(x))
    on_dev = not all(isinstance(a, np.ndarray) for a in (x, y, z, perm))
    mk = (lambda: HArray.empty((N,), f64)) if on_dev else (lambda: np.empty(N, f64))
    xs, ys, zs = mk(), mk(), mk()
    c = Call(x, y, z, perm, xs, ys, zs)
    c.done(_lib.lib().mdh_gather_positions(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), c.inp(perm, i32), N, c.out(xs, f64, upload=False),
                                           c.out(ys, f64, upload=False), c.out(zs, f64, upload=False), c.space, c.stream))
    return xs, ys, zs


def translate_rows(rows_sorted, dist_sorted, counts_sorted, perm):
    """a list built on the sorted copy -> the list of the original order (ids through perm, rows to their atoms' places);
    ``dist_sorted`` / ``counts_sorted`` may be None.  -> (rows, dist, counts)"""
    N, M = int(rows_sorted.shape[0]), int(rows_sorted.shape[1])
    on_dev = not all(a is None or isinstance(a, np.ndarray) for a in (rows_sorted, dist_sorted, counts_sorted, perm))
    mk = (lambda shape, dt: HArray.empty(shape, dt)) if on_dev else (lambda shape, dt: np.empty(shape, dt))
    rows = mk((N, M), i32)
    dist = mk((N, M), f64) if dist_sorted is not None else None
    counts = mk((N,), i32) if counts_sorted is not None else None
    c = Call(rows_sorted, dist_sorted, counts_sorted, perm, rows, dist, counts)
    c.done(_lib.lib().mdh_translate_rows(c.inp(rows_sorted, i32), c.inp(dist_sorted, f64), c.inp(counts_sorted, i32), c.inp(perm, i32), N, M,
                                         c.out(rows, i32, upload=False), None if dist is None else c.out(dist, f64, upload=False),
                                         None if counts is None else c.out(counts, i32, upload=False), c.space, c.stream))
    return rows, dist, counts
