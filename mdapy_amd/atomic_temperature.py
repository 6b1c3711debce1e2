"""Local atomic temperature.  Mirrors ``mdapy.atomic_temperature.AtomicTemperature``
(src/mdapy/atomic_temperature.py:16-140): kinetic temperature of every atom's neighbourhood in its centre-of-mass frame.

Masses come from an ``amass`` column, or from ``element`` through the standard atomic weights below (IUPAC 2013
abridged values — the reference's static table src/mdapy/data.py is outside the hot path; only the elements
listed here are known, anything else needs an explicit ``amass`` column)."""
from __future__ import annotations

import numpy as np

from . import kernels
from .devarray import HArray, as_numpy, empty, have_gpu
from .frame import Frame
from .parallel import get_num_threads

STANDARD_ATOMIC_WEIGHT = {
    "H": 1.008, "He": 4.002602, "Li": 6.94, "Be": 9.0121831, "B": 10.81, "C": 12.011, "N": 14.007, "O": 15.999,
    "F": 18.998403163, "Ne": 20.1797, "Na": 22.98976928, "Mg": 24.305, "Al": 26.9815385, "Si": 28.085,
    "P": 30.973761998, "S": 32.06, "Cl": 35.45, "Ar": 39.948, "K": 39.0983, "Ca": 40.078, "Sc": 44.955908,
    "Ti": 47.867, "V": 50.9415, "Cr": 51.9961, "Mn": 54.938044, "Fe": 55.845, "Co": 58.933194, "Ni": 58.6934,
    "Cu": 63.546, "Zn": 65.38, "Ga": 69.723, "Ge": 72.63, "As": 74.921595, "Se": 78.971, "Br": 79.904, "Kr": 83.798,
    "Rb": 85.4678, "Sr": 87.62, "Y": 88.90584, "Zr": 91.224, "Nb": 92.90637, "Mo": 95.95, "Ru": 101.07,
    "Rh": 102.9055, "Pd": 106.42, "Ag": 107.8682, "Cd": 112.414, "In": 114.818, "Sn": 118.71, "Sb": 121.76,
    "Te": 127.6, "I": 126.90447, "Xe": 131.293, "Cs": 132.90545196, "Ba": 137.327, "La": 138.90547, "Ce": 140.116,
    "Hf": 178.49, "Ta": 180.94788, "W": 183.84, "Re": 186.207, "Os": 190.23, "Ir": 192.217, "Pt": 195.084,
    "Au": 196.966569, "Hg": 200.592, "Tl": 204.38, "Pb": 207.2, "Bi": 208.9804, "Th": 232.0377, "U": 238.02891,
}


class AtomicTemperature:
    def _scaled(self, name):
        """the velocity column in m/s (A/fs x 1e3 x factor, the reference's two multiplications in its order), made where the column
        lives: in HBM from the column's mirror — three host multiplications and 96 MB over PCIe per call at 4 M atoms otherwise"""
        col = self.data[name]
        if have_gpu() and col.dtype == np.float64:
            return HArray((col.device_array().dev() * 1e3) * self.factor)
        return np.ascontiguousarray(as_numpy(col.to_numpy()) * 1e3 * self.factor)

    def __init__(self, data: Frame, verlet_list, distance_list, rc: float, factor: float = 1.0) -> None:
        self.data = data
        self.verlet_list = verlet_list
        self.distance_list = distance_list
        self.rc = rc
        self.factor = factor

    def compute(self) -> None:
        for c in ("vx", "vy", "vz"):
            assert c in self.data.columns, "No velocity information."
        if "amass" in self.data.columns:
            amass = np.ascontiguousarray(self.data["amass"].to_numpy(), dtype=np.float64)
        elif "element" in self.data.columns:
            ele = np.asarray(self.data["element"].to_numpy())
            amass = np.empty(len(ele), np.float64)
            for e in np.unique(ele):
                if str(e) not in STANDARD_ATOMIC_WEIGHT:
                    raise ValueError(f"Unknown element '{e}' in atomic_numbers.")
                amass[ele == e] = STANDARD_ATOMIC_WEIGHT[str(e)]
        else:
            raise ValueError("No atomic mass.")
        self.T = empty(self.data.shape[0], np.float64)
        v = [self._scaled(c) for c in ("vx", "vy", "vz")]
        if "amass" in self.data.columns and self.data["amass"].dtype == np.float64 and have_gpu():
            amass = self.data["amass"]  # (the column's HBM mirror: uploaded once, not with every call)
        kernels.atomtemp.compute_temp(self.verlet_list, self.distance_list, v[0], v[1], v[2], amass, self.T, self.rc,
                               get_num_threads())
