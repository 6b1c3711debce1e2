"""Static structure factor S(k).  Mirrors ``mdapy.structure_factor.StructureFactor`` (src/mdapy/structure_factor.py:170-420):
``mode='debye'`` integrates the radial distribution function (streaming RDF kernel), ``mode='direct'`` sums phases over
the reciprocal-lattice points (sfc.hip).  Partials are Faber-Ziman normalised.  The x-ray / neutron / electron weighted
totals need the reference's form-factor tables (src/mdapy/data.py), which are outside the hot path."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import _sfc
from . import tool_function as tool
from .box import Box
from .devarray import as_numpy
from .frame import Frame
from .parallel import get_num_threads
from .radial_distribution_function import RadialDistributionFunction


class StructureFactor:
    def __init__(self, data: Frame, box: Box, k_min: float, k_max: float, nbins: int, cal_partial: bool = False,
                 atomic_form_factors: bool = False, mode: str = "debye", rc: Optional[float] = None, nbin_rdf: int = 200,
                 window: bool = False) -> None:
        self.data = data
        self.box = box
        self.k_min = float(k_min)
        assert k_min >= 0, "k_min must be non-negative"
        self.k_max = float(k_max)
        assert k_max > k_min, "k_max must be greater than k_min"
        self.nbins = int(nbins)
        assert nbins > 0, "nbins must be positive"
        if atomic_form_factors:
            raise NotImplementedError("form-factor weighted totals need the reference's static tables (not built)")
        self.cal_partial = bool(cal_partial)
        self.mode = mode.lower()
        if self.mode == "rdf":
            self.mode = "debye"
        assert self.mode in ["direct", "debye"], "mode must be 'direct' or 'debye'"
        self.rc = rc
        self.nbin_rdf = int(nbin_rdf)
        self.window = bool(window)
        self.k = self.Sk = self.Sk_partial = None

    def compute(self) -> None:
        for col in ("x", "y", "z"):
            assert col in self.data.columns, f"Column '{col}' must be present"
        if self.mode == "debye":
            self._compute_debye_mode()
        else:
            self._compute_direct_mode()

    @staticmethod
    def _species_labels(view: Frame) -> np.ndarray:
        if "element" in view.columns:
            return np.asarray(view["element"].to_numpy())
        if "type" in view.columns:
            return np.asarray(view["type"].to_numpy())
        return np.zeros(view.shape[0], dtype=np.int32)

    # ---- Debye / RDF method (structure_factor.py:251-330)
    def _compute_debye_mode(self) -> None:
        data, box = self.data, self.box
        L_max = float(max(np.linalg.norm(box.box[i]) for i in range(3)))
        if self.rc is None:
            self.rc = L_max / 2.0
        self.k = np.linspace(self.k_min, self.k_max, self.nbins)
        if self.k_min == 0.0:
            self.k[0] = self.k[1] / 1000.0
        repeat = box.check_small_box(self.rc)
        rep_data, rep_box = data, box
        if sum(repeat) != 3:
            rep_data, rep_box = tool.replicate(data, box, *repeat)
        rx, ry, rz = (np.ascontiguousarray(as_numpy(rep_data[c].to_numpy()), dtype=np.float64) for c in "xyz")
        rdf = RadialDistributionFunction(self.rc, self.nbin_rdf, rep_box, type_list=self._species_labels(rep_data), streaming=True,
                                         x=rx, y=ry, z=rz)
        rdf.compute()
        self._rdf = rdf
        self.r = rdf.r
        elements = list(rdf.elements)
        n_total = rep_data.shape[0]
        rho = n_total / rep_box.volume
        self._uniele = elements
        self._concentrations = np.bincount(rdf.type_list, minlength=len(elements)) / n_total
        self._density = self.num_density = self.density = rho
        w = np.sinc(2.0 * rdf.r / L_max) if self.window else np.ones_like(rdf.r)
        sin_kr = np.sin(np.outer(self.k, rdf.r))
        trapz = getattr(np, "trapezoid", None) or np.trapz
        partial: Dict[Tuple[Any, Any], np.ndarray] = {}
        for a, la in enumerate(elements):
            for lb in elements[a:]:
                g_ab = rdf.g_partial[(la, lb)]
                partial[(la, lb)] = 1.0 + 4.0 * np.pi * rho / self.k * trapz(sin_kr * (rdf.r * (g_ab - 1.0) * w), x=rdf.r, axis=1)
        self.Sk = 1.0 + 4.0 * np.pi * rho / self.k * trapz(sin_kr * (rdf.r * (rdf.g_total - 1.0) * w), x=rdf.r, axis=1)
        if self.cal_partial:
            self.Sk_partial = partial
        else:
            self._Sk_partial_internal = partial

    # ---- direct summation (structure_factor.py:332-420)
    def _compute_direct_mode(self) -> None:
        data, box = self.data, self.box
        edges = np.linspace(self.k_min, self.k_max, self.nbins + 1)
        self.k = (edges[1:] + edges[:-1]) / 2.0
        n = data.shape[0]
        repeat = [1, 1, 1]
        if n < 200 and sum(box.boundary) > 0:
            while np.prod(repeat) * n < 200:
                for i in range(3):
                    if box.boundary[i] == 1:
                        repeat[i] += 1
        if sum(repeat) != 3:
            data, box = tool.replicate(data, box, *repeat)
        if self.cal_partial:
            if "element" in data.columns:
                col = "element"
            elif "type" in data.columns:
                col = "type"
            else:
                raise ValueError("cal_partial / atomic_form_factors require an 'element' or 'type' column.")
            labels = np.asarray(data[col].to_numpy())
            uniele, dense_idx = tool.dense_labels(labels)
        else:
            uniele = ["all"]
        n_total = data.shape[0]
        self._uniele = uniele
        self._density = self.num_density = self.density = n_total / box.volume
        x, y, z = (np.ascontiguousarray(as_numpy(data[c].to_numpy()), dtype=np.float64) for c in "xyz")
        if self.cal_partial:
            type_dense = dense_idx
            c = np.bincount(type_dense, minlength=len(uniele)) / n_total
            self._concentrations = c
            al = np.zeros((len(uniele), len(uniele), self.nbins))
            _sfc.compute_sfc_direct_partial(x, y, z, type_dense, len(uniele), box.box, box.origin, box.boundary, al, self.nbins,
                                            self.k_max, self.k_min, get_num_threads())
            fz: Dict[Tuple[Any, Any], np.ndarray] = {}
            for ia, sa in enumerate(uniele):  # Ashcroft-Langreth -> Faber-Ziman
                for ib in range(ia, len(uniele)):
                    fz[(sa, uniele[ib])] = ((al[ia, ib] - c[ia]) / c[ia] ** 2 + 1.0) if ia == ib else (al[ia, ib] / (c[ia] * c[ib]) + 1.0)
            self.Sk_partial = fz
            self.Sk = al.sum(axis=(0, 1))
        else:
            self.Sk = np.zeros(self.nbins)
            _sfc.compute_sfc_direct(x, y, z, box.box, box.origin, box.boundary, self.Sk, self.nbins, self.k_max, self.k_min,
                                    num_t=get_num_threads())
