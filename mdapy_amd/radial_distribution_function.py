"""Radial distribution function — the drop-in for
``mdapy.radial_distribution_function.RadialDistributionFunction`` (src/mdapy/radial_distribution_function.py:20-211).

The kernels return ordered-pair counts per (species a, species b, shell); this class turns them into ``r``,
``g_total`` and ``g_partial[(label_a, label_b)]`` (a <= b in sorted label order).  Two sources of counts: an existing
cutoff list, or — ``streaming=True`` — the positions themselves through a cell grid of their own, for cutoffs whose list
would not fit in memory."""
import numpy as np

from . import kernels, policy
from .parallel import get_num_threads


class RadialDistributionFunction:
    def __init__(self, rc, nbin, box, verlet_list=None, distance_list=None, neighbor_number=None, type_list=None,
                 streaming=False, x=None, y=None, z=None):
        self.rc, self.nbin = float(rc), int(nbin)
        self.box, self.vol = box, box.volume
        self.streaming = bool(streaming)
        self.verlet_list = self.distance_list = self.neighbor_number = None
        if self.streaming:
            if any(c is None for c in (x, y, z)):
                raise ValueError("streaming=True requires x, y, z position arrays.")
            assert len(x) == len(y) == len(z), "x, y, z must have the same shape"
            self._x, self._y, self._z = x, y, z
            self.N = int(len(x))
        else:
            if any(c is None for c in (verlet_list, distance_list, neighbor_number)):
                raise ValueError("streaming=False requires verlet_list, distance_list, " "neighbor_number.")
            self.verlet_list, self.distance_list, self.neighbor_number = verlet_list, distance_list, neighbor_number
            self.N = int(verlet_list.shape[0])
        labels = np.zeros(self.N, dtype=np.int32) if type_list is None else np.asarray(type_list)
        # (the codes only go to the kernel — and to whoever reads `type_list`, for whom an array in HBM converts itself: streaming, or
        # a list that lives in HBM; the host passes over a 4 M-label column are 19 ms of the first call on a new System)
        in_hbm = self.streaming or type(verlet_list).__name__ in ("HArray", "LazyHArray")
        self.elements, self.type_list = policy.label_codes(labels, device_ok=in_hbm)
        self.Ntype = len(self.elements)

    def _pair_counts(self):
        """ordered-pair counts (species, species, shell) as f64 — integers, exactly"""
        kinds, reach, shells = self.Ntype, self.rc, self.nbin
        counts = np.zeros((kinds, kinds, shells), np.float64)
        lists = (self.verlet_list, self.distance_list, self.neighbor_number)
        if self.streaming:
            where = (self._x, self._y, self._z, self.type_list, *policy.box_args(self.box))
            kernels.rdf._rdf_streaming(*where, counts, reach, shells, get_num_threads())
        elif kinds == 1:
            kernels.rdf._rdf_single_species(*lists, counts[0, 0], reach, shells)
        else:
            kernels.rdf._rdf(*lists, self.type_list, counts, reach, shells)
        return counts

    def compute(self):
        self.r, shell = policy.shell_table(self.rc, self.nbin, self.vol)
        counts = self._pair_counts()
        # all pairs against the ideal-gas expectation N^2 * shell / V
        self.g_total = counts.sum(axis=(0, 1)) / shell / self.N ** 2
        names = self.elements
        population = policy.label_population(self.type_list, len(names))
        both_orders = counts + counts.transpose(1, 0, 2)  # [a, b] + [b, a]
        self.g_partial = {}
        for a, name_a in enumerate(names):
            for b in range(a, len(names)):
                pairs = population[a] * population[b]
                if pairs == 0:
                    curve = np.zeros(self.nbin)
                elif a == b:
                    curve = counts[a, a] / pairs / shell
                else:  # both ordered halves, averaged
                    curve = both_orders[a, b] / pairs / shell * 0.5
                self.g_partial[(name_a, names[b])] = curve
