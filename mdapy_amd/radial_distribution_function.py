"""Radial distribution function.  Mirrors
``mdapy.radial_distribution_function.RadialDistributionFunction``
(src/mdapy/radial_distribution_function.py:20-211): the kernels return ordered-pair
counts; the normalisation to g(r) is done here in numpy exactly as in the reference."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import tool_function as tool

from . import _rdf
from .box import Box
from .parallel import get_num_threads


class RadialDistributionFunction:
    def __init__(self, rc: float, nbin: int, box: Box, verlet_list=None, distance_list=None, neighbor_number=None,
                 type_list=None, streaming: bool = False, x=None, y=None, z=None) -> None:
        self.rc = float(rc)
        self.nbin = int(nbin)
        self.box = box
        self.vol = self.box.volume
        self.streaming = bool(streaming)
        if self.streaming:
            if x is None or y is None or z is None:
                raise ValueError("streaming=True requires x, y, z position arrays.")
            self._x, self._y, self._z = x, y, z
            assert len(x) == len(y) == len(z), "x, y, z must have the same shape"
            self.N = int(len(x))
            self.verlet_list = self.distance_list = self.neighbor_number = None
        else:
            if verlet_list is None or distance_list is None or neighbor_number is None:
                raise ValueError("streaming=False requires verlet_list, distance_list, " "neighbor_number.")
            self.verlet_list = verlet_list
            self.distance_list = distance_list
            self.neighbor_number = neighbor_number
            self.N = int(self.verlet_list.shape[0])
        raw = np.zeros(self.N, dtype=np.int32) if type_list is None else np.asarray(type_list)
        self.elements, self.type_list = tool.dense_labels(raw)  # :125-142 labels -> dense 0..Ntype-1 in sorted order
        self.Ntype = len(self.elements)

    def compute(self) -> None:
        edges = np.linspace(0, self.rc, self.nbin + 1)
        const = (4.0 * np.pi / 3.0 * (edges[1:] ** 3 - edges[:-1] ** 3)) / self.vol
        self.r = (edges[1:] + edges[:-1]) / 2
        counts = np.zeros((self.Ntype, self.Ntype, self.nbin), dtype=np.float64)
        if self.streaming:
            _rdf._rdf_streaming(self._x, self._y, self._z, self.type_list, self.box.box, self.box.origin,
                                self.box.boundary, counts, self.rc, self.nbin, get_num_threads())
        elif self.Ntype > 1:
            _rdf._rdf(self.verlet_list, self.distance_list, self.neighbor_number, self.type_list, counts, self.rc,
                      self.nbin)
        else:
            flat = np.zeros(self.nbin, dtype=np.float64)
            _rdf._rdf_single_species(self.verlet_list, self.distance_list, self.neighbor_number, flat, self.rc,
                                     self.nbin)
            counts[0, 0] = flat
        number_per_type = np.bincount(self.type_list, minlength=self.Ntype)
        total = np.zeros(self.nbin, dtype=np.float64)
        for a in range(self.Ntype):
            for b in range(self.Ntype):
                total += counts[a, b]
        self.g_total = total / const / self.N**2
        self.g_partial: Dict[Tuple[Any, Any], np.ndarray] = {}
        for a in range(self.Ntype):
            n_a = number_per_type[a]
            for b in range(a, self.Ntype):
                n_b = number_per_type[b]
                raw = counts[a, b] if a == b else counts[a, b] + counts[b, a]
                if n_a > 0 and n_b > 0:
                    g_ab = raw / (n_a * n_b) / const
                    if a != b:
                        g_ab *= 0.5
                else:
                    g_ab = np.zeros_like(self.r)
                self.g_partial[(self.elements[a], self.elements[b])] = g_ab
