"""Host-side policy of the hot path, in one place.

What the reference decides in Python, scattered over its analysis classes, is a handful of rules.  They are stated
here once, as functions, and the classes / ``System`` only combine them:

* **thin periodic boxes are replicated** before a search — up to twice the cutoff for a cutoff list
  (src/mdapy/box.py:483-502), up to 15 A for the analyses that take a fixed number of nearest neighbours
  (src/mdapy/common_neighbor_analysis.py:99-109, system.py:2032-2036); replicas are cell-major with the original
  atoms first, so results are read back as the first N rows;
* **an open system too small to have k neighbours** gets the analysis' "nothing found" answer without any search;
* **species labels become dense codes** in sorted label order (radial_distribution_function.py:125-142);
* **pair counts become g(r)** by shell volume and number density (radial_distribution_function.py:147-211).
"""
import numpy as np

from . import kernels
from .box import Box
from .frame import Frame, concat
from .parallel import get_num_threads

NEAREST_SPAN = 15.0  # A: thickness below which a periodic direction is replicated for the k-nearest analyses


def axis_copies(box, span):
    """copies per axis that bring every periodic direction thinner than ``span`` up to it (open directions: 1)"""
    thick = box.get_thickness()
    copies = np.ones(3, dtype=np.int32)
    for axis in range(3):
        if box.boundary[axis] == 1 and thick[axis] < span:
            copies[axis] = int(np.ceil(span / thick[axis]))
    return copies


def is_single(copies):
    return int(copies[0]) == 1 and int(copies[1]) == 1 and int(copies[2]) == 1


def positions(frame):
    """x, y, z columns (frame columns carry their HBM mirror)"""
    return frame["x"], frame["y"], frame["z"]


def box_args(box):
    return box.box, box.origin, box.boundary


def tiled_positions(frame, box, copies):
    """positions of the nx*ny*nz replica (cell-major, originals first; src/repeat_cell.cpp:19-61) and its cell matrix"""
    nx, ny, nz = (int(c) for c in copies)
    source = np.ascontiguousarray(frame.select("x", "y", "z").to_numpy(), dtype=np.float64)
    flat = np.zeros(source.shape[0] * nx * ny * nz * 3, dtype=np.float64)
    kernels.repeat_cell.repeat_cell(flat, box.box, source, nx, ny, nz, get_num_threads())
    return flat.reshape((-1, 3)), box.box * np.array([[nx], [ny], [nz]])


def replica(frame, box, copies, all_columns):
    """(frame, box) of the replicated system.  ``all_columns``: carry every per-atom column along (an ``id`` column is
    renumbered from 1); otherwise positions only — enough for a structure analysis that reads nothing else."""
    pos, cell = tiled_positions(frame, box, copies)
    if all_columns:
        n = int(copies[0]) * int(copies[1]) * int(copies[2])
        out = concat([frame] * n).with_columns(x=pos[:, 0], y=pos[:, 1], z=pos[:, 2])
        if "id" in out.columns:
            out = out.with_columns(id=np.arange(1, out.shape[0] + 1, dtype=np.asarray(frame["id"]).dtype))
    else:
        out = Frame({"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]})
    return out, Box(cell, box.boundary, box.origin)


def widened(frame, box, span, all_columns=False):
    """(frame, box, replicated?) with every periodic direction at least ``span`` thick"""
    copies = axis_copies(box, span)
    if is_single(copies):
        return frame, box, False
    grown, grown_box = replica(frame, box, copies, all_columns)
    return grown, grown_box, True


def hopeless(box, natoms, k):
    """an open system of at most k atoms: no atom has k neighbours, the analyses answer 'other' without a search"""
    return int(np.sum(box.boundary)) == 0 and natoms <= k


def nearest_rows(frame, box, k):
    """rows of the k nearest neighbours of every atom, searched here (the caller had no list to offer)"""
    from .knn import NearestNeighbor

    finder = NearestNeighbor(frame, box, k)
    finder.compute()
    return finder.indices_py


import weakref

_label_cache = []  # [(weak reference to a read-only label array, names, codes)]: frame columns are immutable, analyses re-ask


def label_codes(labels, device_ok=False):
    """(sorted distinct labels as Python objects, int32 code of every atom in that order)

    device_ok: the caller hands the codes to a kernel and reads them through label_population at most — a large int32 column is
    then coded in HBM (mdh_dense_codes_i32: 0.3 ms for 10 M labels where the numpy passes below take 65 ms — the whole difference
    between the first partial-RDF call on a new System and the following ones) and the codes come back as an HArray."""
    raw = np.asarray(labels)
    if raw.size == 0:
        return [], np.zeros(0, np.int32)
    from .devarray import HArray, _cache_lock, frozen_by_us, mark_frozen

    if frozen_by_us(raw):  # a column the package froze itself: nobody else's array is trusted to stay what it was
        with _cache_lock:
            for ref, names, codes in _label_cache:
                if ref() is raw and (device_ok or not isinstance(codes, HArray)):
                    return list(names), codes
        made = _label_codes_device(raw) if device_ok else None
        if made is not None:
            names, codes, counts = made
            _population[id(codes)] = (weakref.ref(codes), counts)
        else:
            names, codes = _label_codes(raw)
            codes.setflags(write=False)  # shared by every later call on the same column (and mirrored in HBM once, devarray._mirror_of)
            mark_frozen(codes)
        try:
            with _cache_lock:
                _label_cache.append((weakref.ref(raw), names, codes))
                del _label_cache[:-4]
        except TypeError:
            pass
        return list(names), codes
    made = _label_codes_device(raw) if device_ok else None
    if made is not None:
        if len(_population) > 8:  # (entries of arrays that are gone: a trajectory loop makes one per frame)
            for k in [k for k, (ref, _) in _population.items() if ref() is None]:
                del _population[k]
        _population[id(made[1])] = (weakref.ref(made[1]), made[2])
        return made[0], made[1]
    return _label_codes(raw)


def _label_codes_device(raw):
    """(names, codes in HBM, atoms per code) of a large int32 label column, or None when the host passes have to do it"""
    from . import _lib, devarray

    if raw.dtype != np.int32 or raw.ndim != 1 or raw.size < (1 << 16) or not devarray.have_gpu():
        return None
    held = devarray._mirror_of(np.ascontiguousarray(raw))
    low, high = held._min_max_i32()
    span = high - low + 1
    if span > 4096:
        return None
    counts = np.zeros(span, np.int64)
    codes = devarray.HArray.empty((raw.shape[0],), np.int32)
    _lib.check(_lib.lib().mdh_dense_codes_i32(held.data_ptr(), int(raw.shape[0]), int(low), int(span), codes.data_ptr(), counts.ctypes.data,
                                              _lib.DEVICE, devarray.current_stream_ptr()))
    present = np.flatnonzero(counts)
    return (present + low).tolist(), codes, counts[present]


def label_population(codes, kinds):
    """atoms per species code; cached with the codes of an immutable column"""
    hit = _population.get(id(codes))
    if hit is not None and hit[0]() is codes:
        pop = np.asarray(hit[1])
        return pop if len(pop) >= kinds else np.concatenate([pop, np.zeros(kinds - len(pop), pop.dtype)])
    from .devarray import HArray

    if isinstance(codes, HArray):
        codes = codes.numpy()
    for ref, names, cached in list(_label_cache):
        if cached is codes:
            if len(_population) > 8:
                _population.clear()
            key = id(cached)
            if key not in _population or _population[key][0]() is not cached:
                _population[key] = (weakref.ref(cached), np.bincount(cached, minlength=kinds))
            return _population[key][1]
    return np.bincount(codes, minlength=kinds)


_population = {}


def _label_codes(raw):
    head = raw.flat[0]
    if bool((raw == head).all()):  # one species: no sort needed
        return [head.item() if hasattr(head, "item") else head], np.zeros(raw.shape[0], np.int32)
    if raw.dtype.kind in "iu":  # numeric types: a counting pass instead of a sort of N labels
        low, high = int(raw.min()), int(raw.max())
        if high - low < (1 << 20):
            shifted = raw - low
            present = np.flatnonzero(np.bincount(shifted.reshape(-1), minlength=high - low + 1))
            lut = np.zeros(high - low + 1, np.int32)
            lut[present] = np.arange(len(present), dtype=np.int32)
            return (present + low).tolist(), lut[shifted.reshape(-1)]
    elif raw.dtype == object:  # element symbols: hash them (pandas) rather than compare Python strings N log N times
        try:
            import pandas as pd

            codes, names = pd.factorize(raw.reshape(-1), sort=True)
            return list(names), codes.astype(np.int32)
        except ImportError:
            pass
    names, codes = np.unique(raw, return_inverse=True)
    return names.tolist(), codes.reshape(-1).astype(np.int32)


def species_of(frame):
    """per-atom species labels of a frame: elements if it has them, numeric types otherwise, one species if neither"""
    for column in ("element", "type"):
        if column in frame.columns:
            return frame[column].to_numpy()
    return np.zeros(frame.shape[0], np.int32)


def shell_table(rc, nbin, volume):
    """(bin centres, shell volume / system volume per bin) of nbin equal-width shells out to rc"""
    edge = np.linspace(0, rc, nbin + 1)
    outer, inner = edge[1:], edge[:-1]
    return (outer + inner) / 2, (4.0 * np.pi / 3.0 * (outer ** 3 - inner ** 3)) / volume
