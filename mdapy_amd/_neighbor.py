"""Drop-in for the reference's ``mdapy._neighbor`` nanobind module
(src/neighbor.cpp:841-859): same function names, argument order and
caller-allocated outputs, executed by the HIP kernels in csrc/neighbor.hip.
Arrays may be numpy (host; staged by the library) or HBM resident
(:class:`mdapy_amd.devarray.HArray`, frame columns, torch ROCm tensors)."""
import numpy as np

from . import _lib
from .devarray import Call, HArray, have_gpu

f64, i32 = np.float64, np.int32


def build_neighbor(x, y, z, box, origin, boundary, rc, verlet_list, distance_list, neighbor_number, num_t=1,
                   fill_pads=False, key=None):
    """src/neighbor.cpp:351.  Extensions: ``fill_pads`` lets the kernel write the -1 / rc+1 pads so the caller may pass
    uninitialised buffers; ``key`` (i64, N) orders the atoms of a cell by descending key instead of descending index (the
    global ids of a slab's owned + ghost atoms: rows then equal those of the undivided system)."""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, distance_list, neighbor_number, key)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_build_neighbor_keyed(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, float(rc),
                                              c.out(verlet_list, i32, upload=not fill_pads),
                                              c.out(distance_list, f64, upload=not fill_pads),
                                              c.out(neighbor_number, i32, upload=False), M, int(bool(fill_pads)),
                                              c.inp(key, np.int64), c.space, c.stream)
    c.done(rc_)


def hint_centre_window(axis, frac_lo, frac_hi):
    """with hint_cell_window: the stretch of the window that holds the atoms whose rows are wanted (a rank's own slab) — the next
    build makes no rows for the ghosts around it (mdh_hint_centre_window); pre-zero the counts"""
    _lib.check(_lib.lib().mdh_hint_centre_window(int(axis), float(frac_lo), float(frac_hi)))


def hint_cell_window(axis, frac_lo, frac_hi):
    """promise to the next ``build_neighbor`` of this thread: every atom's wrapped fractional coordinate along ``axis`` lies in
    [frac_lo, frac_hi] (a rank's slab and halo in the global box): the passes over all cells of the global grid run over that
    window's planes only (mdh_hint_cell_window).  Same results."""
    _lib.check(_lib.lib().mdh_hint_cell_window(int(axis), float(frac_lo), float(frac_hi)))


def cell_window_check(stream=None):
    """wait for the stream and raise ValueError if the last windowed build of this thread left atoms out (broken promise);
    without this the NEXT build of the thread raises"""
    if stream is None:
        from .devarray import torch

        stream = int(torch().cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().mdh_cell_window_check(stream))


def build_neighbor_fcna(x, y, z, box, origin, boundary, rc, verlet_list, distance_list, neighbor_number, pattern, num_t=1,
                        fill_pads=False, key=None):
    """``build_neighbor`` (src/neighbor.cpp:351) and ``fcna`` (src/cna.cpp:429) with the same ``rc`` in one pass over the
    tiles: the lists as ``build_neighbor`` leaves them, ``pattern`` (caller-initialised) as ``fcna`` leaves it; ``key`` as
    in ``build_neighbor``."""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z, verlet_list, distance_list, neighbor_number, pattern, key)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_build_neighbor_fcna(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, float(rc),
                                             c.out(verlet_list, i32, upload=not fill_pads),
                                             c.out(distance_list, f64, upload=not fill_pads),
                                             c.out(neighbor_number, i32, upload=False), M, int(bool(fill_pads)),
                                             c.out(pattern, i32), c.inp(key, np.int64), c.space, c.stream)
    c.done(rc_)


def build_neighbor_without_max_neigh(x, y, z, box, origin, boundary, rc, num_t=1, key=None, pattern=None):
    """src/neighbor.cpp:189: exact row width = max neighbour count (>= 1); returns (verlet, dist, nn).
    One library call (mdh_build_neighbor_exact): the cell grid is built once for the counting pass and the build; the rows
    are allocated between the two through a callback, as the reference allocates them inside the call (:312-317).
    The arrays are HBM resident when the inputs are (or when a GPU is present and inputs are frame columns).
    ``key`` (i64, N): as in ``build_neighbor`` (mdh_build_neighbor_exact_keyed).
    ``pattern`` (i32, N, caller-initialised; extension): the fixed-cutoff CNA labels of this cutoff as ``_cna.fcna`` would leave
    them on the finished lists, written in the same pass over the tiles (mdh_build_neighbor_exact_fcna)."""
    import ctypes

    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    N = int(len(x))
    on_dev = not all(isinstance(a, np.ndarray) for a in (x, y, z))
    nn = HArray.empty((N,), i32) if on_dev else np.zeros(N, i32)
    rows = {}

    @_lib.ALLOC_ROWS
    def alloc(user, n, m, pv, pd):
        try:
            if on_dev:
                v, d = HArray.empty((int(n), int(m)), i32), HArray.empty((int(n), int(m)), f64)
                pv[0] = ctypes.cast(v.data_ptr(), ctypes.POINTER(ctypes.c_int))
                pd[0] = ctypes.cast(d.data_ptr(), ctypes.POINTER(ctypes.c_double))
            else:
                v, d = np.empty((int(n), int(m)), i32), np.empty((int(n), int(m)), f64)
                pv[0] = v.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
                pd[0] = d.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            rows["v"], rows["d"] = v, d
            rows.setdefault("all", []).append((v, d))  # a first allocation at a hinted width may be discarded: alive until the call returns
            return 0
        except Exception:  # an exception must not cross the C frame: the library reports the failed allocation
            return 1

    width = ctypes.c_int64(0)
    c = Call(x, y, z, nn, key, pattern)
    rc_ = _lib.lib().mdh_build_neighbor_exact_fcna(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), N, pb, po, pp, float(rc),
                                                   c.out(nn, i32, upload=False), ctypes.addressof(width), alloc, None,
                                                   c.out(pattern, i32) if pattern is not None else None,
                                                   c.inp(key, np.int64), c.space, c.stream)
    c.done(rc_)
    return rows["v"], rows["d"], nn


def sort_verlet_by_distance(verlet_list, distance_list, sortNum, num_t=1):
    """src/neighbor.cpp:745"""
    c = Call(verlet_list, distance_list)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_sort_verlet_by_distance(c.out(verlet_list, i32), c.out(distance_list, f64), N, M,
                                                 int(sortNum), c.space, c.stream)
    c.done(rc_)


def wrap_positions(x, y, z, box, origin, boundary, num_t=1):
    """src/neighbor.cpp:675 (in place)"""
    keep, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    c = Call(x, y, z)
    rc_ = _lib.lib().mdh_wrap_positions(c.out(x, f64), c.out(y, f64), c.out(z, f64), int(x.shape[0]), pb, po, pp,
                                        c.space, c.stream)
    c.done(rc_)


def average_by_neighbor(rc, verlet_list, distance_list, neighbor_number, value, value_ave, include_self, num_t=1):
    """src/neighbor.cpp:704"""
    c = Call(verlet_list, distance_list, neighbor_number, value, value_ave)
    N, M = int(verlet_list.shape[0]), int(verlet_list.shape[1])
    rc_ = _lib.lib().mdh_average_by_neighbor(float(rc), c.inp(verlet_list, i32), c.inp(distance_list, f64),
                                             c.inp(neighbor_number, i32), N, M, c.inp(value, f64),
                                             c.out(value_ave, f64, upload=False), int(bool(include_self)), c.space,
                                             c.stream)
    c.done(rc_)


def filter_overlap_atom(x, y, z, box, origin, boundary, rc, num_t=1):
    """src/neighbor.cpp:390 — bool array: False for every atom that has a lower-numbered atom within rc"""
    keep_, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    n = int(len(x))
    out = np.zeros(n, np.uint8)
    c = Call(x, y, z)
    if c.space == _lib.DEVICE:
        from .devarray import HArray
        h = HArray.empty(n, np.uint8)
        rc_ = _lib.lib().mdh_filter_overlap_atom(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, float(rc),
                                                 h.data_ptr(), c.space, c.stream)
        c.done(rc_)
        return h.dev().cpu().numpy().astype(bool)
    rc_ = _lib.lib().mdh_filter_overlap_atom(c.inp(x, f64), c.inp(y, f64), c.inp(z, f64), n, pb, po, pp, float(rc),
                                             out.ctypes.data, c.space, c.stream)
    c.done(rc_)
    return out.astype(bool)


def filter_overlap_atom_with_grain(x, y, z, type_list, grain_id, box, origin, boundary, rc_metal_metal, rc_cc, rc_metal_c, num_t=1):
    """src/neighbor.cpp:489 — bool array of the atoms that stay (type 1 metal / 2 carbon, grain ids); the reference's sweep as it
    runs with one thread (with several its result depends on the schedule)"""
    keep_, (pb, po, pp) = _lib.host_box(box, origin, boundary)
    n = int(len(x))
    _lib.same_rows("filter_overlap_atom_with_grain", n, y=y, z=z, type=type_list, grain_id=grain_id)
    xs, ys, zs = (np.ascontiguousarray(np.asarray(a, f64)) for a in (x, y, z))
    t = np.ascontiguousarray(np.asarray(type_list), dtype=i32)
    g = np.ascontiguousarray(np.asarray(grain_id), dtype=i32)
    out = np.zeros(n, np.uint8)
    _lib.check(_lib.lib().mdh_filter_overlap_atom_with_grain(xs.ctypes.data, ys.ctypes.data, zs.ctypes.data, t.ctypes.data, g.ctypes.data, n,
                                                             pb, po, pp, float(rc_metal_metal), float(rc_cc), float(rc_metal_c),
                                                             out.ctypes.data, _lib.HOST, None))
    return out.astype(bool)
