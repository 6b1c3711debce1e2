"""Static structure factor S(k) — the drop-in for ``mdapy.structure_factor.StructureFactor``
(src/mdapy/structure_factor.py:170-655).

Two routes to the same quantity: ``mode='debye'`` Fourier-transforms the radial distribution function (counted by the
streaming RDF kernel), ``mode='direct'`` sums phases over the reciprocal-lattice points (csrc/sfc.hip).  ``Sk`` is the
total, ``Sk_partial[(a, b)]`` the Faber-Ziman partials (a <= b in sorted label order, each tends to 1 at large k).
``atomic_form_factors=True`` — or a later call of ``get_xray / get_neutron / get_electron_structure_factor`` — weights
the partials with the scattering factors of :mod:`mdapy_amd.scattering`:

    S_w(k) = sum_ab (2 - delta_ab) c_a c_b f_a f_b A_ab(k) / (sum_a c_a f_a)^2
"""
import numpy as np

from . import kernels, policy, scattering
from .devarray import as_numpy
from .parallel import get_num_threads
from .radial_distribution_function import RadialDistributionFunction

_integrate = getattr(np, "trapezoid", None) or np.trapz
_DIRECT_MIN_ATOMS = 200  # a smaller periodic system is replicated first: too few reciprocal points per bin otherwise
_WEIGHTS = {"xray": scattering.xray, "neutron": scattering.neutron, "electron": scattering.electron}


def _coordinates(frame):
    return tuple(np.ascontiguousarray(as_numpy(frame[c].to_numpy()), dtype=np.float64) for c in ("x", "y", "z"))


class StructureFactor:
    def __init__(self, data, box, k_min, k_max, nbins, cal_partial=False, atomic_form_factors=False, mode="debye", rc=None,
                 nbin_rdf=200, window=False):
        if not k_min >= 0:
            raise AssertionError("k_min must be non-negative")
        if not k_max > k_min:
            raise AssertionError("k_max must be greater than k_min")
        if not nbins > 0:
            raise AssertionError("nbins must be positive")
        self.data, self.box = data, box
        self.k_min, self.k_max, self.nbins = float(k_min), float(k_max), int(nbins)
        self.atomic_form_factors = bool(atomic_form_factors)
        self.cal_partial = bool(cal_partial) or self.atomic_form_factors  # the weighted totals are built from the partials
        self.mode = {"rdf": "debye"}.get(mode.lower(), mode.lower())
        if self.mode not in ("direct", "debye"):
            raise AssertionError("mode must be 'direct' or 'debye'")
        self.rc, self.nbin_rdf, self.window = rc, int(nbin_rdf), bool(window)
        self.k = self.Sk = self.Sk_partial = None
        self.Sk_xray = self.Sk_neutron = self.Sk_electron = None
        self._uniele = self._concentrations = self._density = None

    def compute(self):
        for name in ("x", "y", "z"):
            if name not in self.data.columns:
                raise AssertionError(f"Column '{name}' must be present")
        (self._compute_debye_mode if self.mode == "debye" else self._compute_direct_mode)()
        if self.atomic_form_factors:
            self.get_xray_structure_factor()

    def _set_density(self, atoms, volume):
        self._density = self.num_density = self.density = atoms / volume

    # ------------------------------------------------------------- Debye route
    def _compute_debye_mode(self):
        frame, cell = self.data, self.box
        longest = float(max(np.linalg.norm(edge) for edge in cell.box))
        if self.rc is None:
            self.rc = longest / 2.0
        self.k = np.linspace(self.k_min, self.k_max, self.nbins)
        if self.k_min == 0.0:
            self.k[0] = self.k[1] / 1000.0  # the transform divides by k
        copies = cell.check_small_box(self.rc)
        if not policy.is_single(copies):
            frame, cell = policy.replica(frame, cell, copies, all_columns=True)
        x, y, z = _coordinates(frame)
        pairs = RadialDistributionFunction(self.rc, self.nbin_rdf, cell, type_list=policy.species_of(frame), streaming=True,
                                           x=x, y=y, z=z)
        pairs.compute()
        self._rdf, self.r = pairs, pairs.r
        atoms = frame.shape[0]
        self._set_density(atoms, cell.volume)
        self._uniele = list(pairs.elements)
        self._concentrations = policy.label_population(pairs.type_list, len(self._uniele)) / atoms
        damp = np.sinc(2.0 * pairs.r / longest) if self.window else np.ones_like(pairs.r)
        phase = np.sin(np.outer(self.k, pairs.r))
        scale = 4.0 * np.pi * self._density / self.k

        def transform(g):  # 1 + 4 pi rho / k * integral of r (g - 1) sin(k r) dr
            return 1.0 + scale * _integrate(phase * (pairs.r * (g - 1.0) * damp), x=pairs.r, axis=1)

        partial = {pair: transform(g) for pair, g in pairs.g_partial.items()}
        self.Sk = transform(pairs.g_total)
        if self.cal_partial:
            self.Sk_partial = partial
        else:
            self._Sk_partial_internal = partial

    # ------------------------------------------------------------ direct route
    def _compute_direct_mode(self):
        frame, cell = self.data, self.box
        edge = np.linspace(self.k_min, self.k_max, self.nbins + 1)
        self.k = (edge[1:] + edge[:-1]) / 2.0
        atoms = frame.shape[0]
        periodic = [a for a in range(3) if cell.boundary[a] == 1]
        copies = [1, 1, 1]
        while periodic and atoms * copies[0] * copies[1] * copies[2] < _DIRECT_MIN_ATOMS:
            for a in periodic:
                copies[a] += 1
        if not policy.is_single(copies):
            frame, cell = policy.replica(frame, cell, copies, all_columns=True)
            atoms = frame.shape[0]
        self._set_density(atoms, cell.volume)
        where = (*_coordinates(frame),)
        if not self.cal_partial:
            self._uniele = ["all"]
            self.Sk = np.zeros(self.nbins)
            kernels.sfc.compute_sfc_direct(*where, *policy.box_args(cell), self.Sk, self.nbins, self.k_max, self.k_min,
                                           num_t=get_num_threads())
            return
        column = next((c for c in ("element", "type") if c in frame.columns), None)
        if column is None:
            raise ValueError("cal_partial / atomic_form_factors require an 'element' or 'type' column.")
        names, codes = policy.label_codes(np.asarray(frame[column].to_numpy()))
        kinds = len(names)
        share = np.bincount(codes, minlength=kinds) / atoms
        self._uniele, self._concentrations = names, share
        al = np.zeros((kinds, kinds, self.nbins))  # Ashcroft-Langreth partials
        kernels.sfc.compute_sfc_direct_partial(*where, codes, kinds, *policy.box_args(cell), al, self.nbins, self.k_max,
                                               self.k_min, get_num_threads())
        self.Sk = al.sum(axis=(0, 1))
        # Ashcroft-Langreth -> Faber-Ziman
        self.Sk_partial = {}
        for a in range(kinds):
            for b in range(a, kinds):
                if a == b:
                    self.Sk_partial[(names[a], names[b])] = (al[a, a] - share[a]) / share[a] ** 2 + 1.0
                else:
                    self.Sk_partial[(names[a], names[b])] = al[a, b] / (share[a] * share[b]) + 1.0

    # --------------------------------------------------- form-factor weighting
    def _weighted_total(self, kind):
        partial = self.Sk_partial if self.Sk_partial is not None else getattr(self, "_Sk_partial_internal", None)
        if partial is None:
            raise RuntimeError("Run compute() with cal_partial=True (or set " "atomic_form_factors=True) before requesting a "
                               f"{kind}-weighted total.")
        if kind not in _WEIGHTS:
            raise ValueError(f"unknown weighting kind: {kind!r}")
        names, share = self._uniele, self._concentrations
        factor = [_WEIGHTS[kind](name, self.k) for name in names]
        mean = sum(c * f for c, f in zip(share, factor))
        total = 0.0
        for (a, b), curve in partial.items():
            ia, ib = names.index(a), names.index(b)
            total = total + (1.0 if ia == ib else 2.0) * share[ia] * share[ib] * factor[ia] * factor[ib] * curve
        result = total / mean ** 2
        if kind == "neutron" and np.iscomplexobj(result):  # absorbing isotopes: the modulus
            return np.real(result * np.conj(result)) ** 0.5
        return result

    def get_xray_structure_factor(self):
        self.Sk_xray = self._weighted_total("xray")
        return self.Sk_xray

    def get_neutron_structure_factor(self):
        self.Sk_neutron = self._weighted_total("neutron")
        return self.Sk_neutron

    def get_electron_structure_factor(self):
        self.Sk_electron = self._weighted_total("electron")
        return self.Sk_electron

    # ------------------------------------------------------- back to real space
    def _real_space(self, kind, r=None):
        """(r, g, G, R) from a weighted total S(k): G(r) = 2/pi * integral of k (S - 1) sin(k r) dk (trapezoid rule over the k
        points), g = 1 + G / (4 pi r rho) with g(0) = 0, R = 4 pi r^2 rho g   (src/mdapy/structure_factor.py:560-588)"""
        if kind not in _WEIGHTS:
            raise ValueError(f"unknown weighting kind: {kind!r}")
        total = getattr(self, f"get_{kind}_structure_factor")()
        if r is None:
            r = self.r if hasattr(self, "r") else np.linspace(0.0, np.pi / (self.k[1] - self.k[0]), 200)
        r = np.asarray(r, dtype=float)
        rho = self._density
        reduced = (2.0 / np.pi) * np.trapezoid(np.sin(np.outer(r, self.k)) * self.k * (total - 1.0), x=self.k, axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            pair = np.where(r > 0, reduced / (4.0 * np.pi * r * rho) + 1.0, 0.0)
        return r, pair, reduced, 4.0 * np.pi * r ** 2 * rho * pair


def _real_space_getter(kind, pick, what):
    def getter(self, r=None):
        out = self._real_space(kind, r)
        return out[0], out[pick]

    getter.__doc__ = f"(r, {what}) reconstructed from the {kind}-weighted total S(k)"
    return getter


for _kind in ("xray", "neutron", "electron"):  # the nine accessors of the reference (structure_factor.py:590-653)
    setattr(StructureFactor, f"get_{_kind}_pair_distribution_function", _real_space_getter(_kind, 1, "g(r)"))
    setattr(StructureFactor, f"get_{_kind}_reduced_pair_distribution_function", _real_space_getter(_kind, 2, "G(r) = 4 pi r rho (g - 1)"))
    setattr(StructureFactor, f"get_{_kind}_radial_distribution_function", _real_space_getter(_kind, 3, "R(r) = 4 pi r^2 rho g"))
del _kind
