"""k nearest neighbours — the drop-in for ``mdapy.knn.NearestNeighbor`` (src/mdapy/knn.py:17-129).

``compute()`` leaves ``indices_py`` (N, k) int32 and ``distances_py`` (N, k) f64, nearest first; periodic images count
as separate neighbours.  Only a system with fewer than k atoms is replicated (three more copies per periodic axis
until k atoms exist), and then ``_enlarge_data`` / ``_enlarge_box`` describe what the indices refer to."""
import numpy as np

from . import kernels, policy
from .devarray import empty
from .parallel import get_num_threads

MAX_K = 24  # the search keeps its candidates in a fixed-size list (src/mdapy/knn.py:14)

import weakref

_bags = weakref.WeakKeyDictionary()  # the x column of a frame -> (weak y column, weak z column, dict of candidate rows)


def _candidates_of(frame):
    """the dict that holds the candidate rows of searches over THESE position columns (column objects are immutable and shared by
    the frames a System makes of them — a new per-atom column does not lose the rows; new positions are new columns)"""
    try:
        xc, yc, zc = frame["x"], frame["y"], frame["z"]
        hit = _bags.get(xc)
        if hit is None or hit[0]() is not yc or hit[1]() is not zc:
            hit = _bags[xc] = (weakref.ref(yc), weakref.ref(zc), {})
        return hit[2]
    except TypeError:  # (a column type that cannot be weakly referenced: no sharing)
        return None


class NearestNeighbor:
    def __init__(self, data, box, k):
        for name in ("x", "y", "z"):
            if name not in data.columns:
                raise AssertionError(f"data must contain column {name!r}.")
        if data.shape[0] <= 0:
            raise AssertionError("data must contain at least one atom.")
        self.k = int(k)
        if not (1 <= self.k <= MAX_K):
            raise AssertionError(f"k must be in [1, {MAX_K}], got {self.k}.")
        self.data, self.box = data, box

    def _copies_for_k(self):
        """copies per axis so that the searched system holds at least k atoms"""
        copies = [1, 1, 1]
        atoms = self.data.shape[0]
        if atoms >= self.k:
            return copies
        periodic = [a for a in range(3) if self.box.boundary[a] == 1]
        if not periodic:
            raise AssertionError(
                f"Need periodic boundary if you want to query {self.k} neighbors " f"in {atoms}-atom system."
            )
        while atoms * copies[0] * copies[1] * copies[2] < self.k:
            for a in periodic:
                copies[a] += 3
        return copies

    # (name kept: part of the reference class' surface)
    def _check_repeat_nearest(self):
        return self._copies_for_k()

    def compute(self):
        frame, cell = self.data, self.box
        copies = self._copies_for_k()
        if not policy.is_single(copies):
            frame, cell = policy.replica(frame, cell, copies, all_columns=True)
            self._enlarge_data, self._enlarge_box = frame, cell
        rows = frame.shape[0]
        self.indices_py = empty((rows, self.k), np.int32)
        self.distances_py = empty((rows, self.k), np.float64)
        # a frame that is a cell-sorted copy of another (System's twin) carries the original index of its atoms: exact ties in
        # distance are then ordered as they are in the original system (which neighbours of a perfect lattice are listed depends on it)
        key = getattr(frame, "order_key", None) if frame is self.data else None
        extra = {} if key is None else {"key": key}
        # the candidate rows of the search's cutoff build stay with the FRAME (its columns never change, frame.py): the next
        # search of the same frame in the same box — another k, another analysis — skips that build
        if frame is self.data and getattr(kernels.fast_knn, "keeps_candidates", False):
            bag = _candidates_of(frame)
            if bag is not None:
                extra["candidates"] = bag
        kernels.fast_knn.knn(*policy.positions(frame), *policy.box_args(cell), self.k, self.indices_py, self.distances_py,
                             get_num_threads(), **extra)
