"""Cutoff neighbor list — the drop-in for ``mdapy.neighbor.Neighbor`` (src/mdapy/neighbor.py:13-142).

Same constructor, checks and error texts; ``compute()`` leaves ``verlet_list`` (N, M) int32 padded with -1,
``distance_list`` (N, M) f64 padded with rc + 1 and ``neighbor_number`` (N) int32, plus ``_enlarge_data`` /
``_enlarge_box`` when a thin periodic box had to be replicated first.  With ``max_neigh=None`` M is the largest
count found (one library call, one cell grid for counting and building); with a fixed ``max_neigh`` an overflow is a
``ValueError`` that names the width that would have fitted."""
import numpy as np

from . import kernels, policy
from .devarray import empty, zeros
from .parallel import get_num_threads

_NEEDS = ("x", "y", "z")


class Neighbor:
    def __init__(self, rc, box, data, max_neigh=None, key=None):
        # key (extension; int64 per atom): in-cell ordering key of the build — the original index of every atom of a cell-sorted
        # copy (System's twin), so that its rows come out in the order the original system's rows have
        self.key = key
        self.rc = float(rc)
        if not self.rc > 0:
            raise AssertionError(f"rc must be positive, got {self.rc}.")
        self.max_neigh = None if max_neigh is None else int(max_neigh)
        if self.max_neigh is not None and self.max_neigh <= 0:
            raise AssertionError(f"max_neigh must be positive, got {self.max_neigh}.")
        missing = [name for name in _NEEDS if name not in data.columns]
        if missing:
            raise AssertionError(f"data must contain column {missing[0]!r}.")
        self.box, self.data = box, data
        self.N = data.shape[0]
        if self.N <= 0:
            raise AssertionError("data must contain at least one atom.")

    def compute(self, label=False):
        """``label`` (extension): leave ``pattern`` as well — the fixed-cutoff common-neighbour labels of this cutoff, what
        ``_cna.fcna`` makes of the finished lists (src/cna.cpp:429-506), written in the same pass over the tiles (a centre's 12
        or 14 neighbours are still staged in LDS when its row is written).  ``None`` where the search ran on a replica."""
        # a periodic direction thinner than two cutoffs would make an atom its own neighbour's image: search a replica
        frame, cell, grown = policy.widened(self.data, self.box, 2.0 * self.rc, all_columns=True)
        if grown:
            self._enlarge_data, self._enlarge_box = frame, cell
        where = (*policy.positions(frame), *policy.box_args(cell), self.rc)
        if self.key is not None and grown:
            raise AssertionError("an ordering key cannot follow a system into its replica")
        self._key = {} if self.key is None else {"key": self.key}
        self.pattern = zeros(frame.shape[0], np.int32) if (label and not grown) else None  # (the kernels only ever raise a label)
        if self.max_neigh is None:
            extra = dict(self._key) if self.pattern is None else dict(self._key, pattern=self.pattern)
            rows = kernels.neighbor.build_neighbor_without_max_neigh(*where, get_num_threads(), **extra)
            self.verlet_list, self.distance_list, self.neighbor_number = rows
            return
        width, atoms = self.max_neigh, frame.shape[0]
        self.verlet_list = empty((atoms, width), np.int32)
        self.distance_list = empty((atoms, width), np.float64)
        self.neighbor_number = zeros((atoms,), np.int32)  # (an atom with a NaN coordinate takes no cell and gets no row: its count reads 0)
        self._search_fixed(where)
        longest = int(self.neighbor_number.max(initial=0))
        if longest > width:
            raise ValueError(
                f"max_neigh={width} is too small: at least one "
                f"atom has {longest} neighbors within rc={self.rc}. "
                f"Re-run with max_neigh>={longest} (or omit max_neigh "
                "to let mdapy size the buffer automatically)."
            )

    def _search_fixed(self, where):
        out = (self.verlet_list, self.distance_list, self.neighbor_number)
        if self.pattern is None:
            build = kernels.neighbor.build_neighbor
        else:
            def build(*a, **kw):  # lists and labels in one pass (mdh_build_neighbor_fcna)
                n = len(where) + len(out)
                kernels.neighbor.build_neighbor_fcna(*a[:n], self.pattern, *a[n:], **kw)
        if isinstance(self.verlet_list, np.ndarray):
            # host buffers: the caller pre-fills the pads, as the reference's Python does (neighbor.py:125-129)
            self.verlet_list[...] = -1
            self.distance_list[...] = self.rc + 1.0
            self.neighbor_number[...] = 0
            build(*where, *out, get_num_threads(), **self._key)
        else:  # HBM buffers: the kernel writes the pads itself, no extra pass over 12 M bytes per atom
            build(*where, *out, get_num_threads(), fill_pads=True, **self._key)
