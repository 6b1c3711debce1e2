"""Simulation box (host side).  Mirrors the public surface of the reference's
``mdapy.box.Box`` (src/mdapy/box.py:93-502): row-vector 3x3 matrix, origin,
0/1 boundary flags, perpendicular thickness and the small-box replication rule.
The kernels rebuild their own device-side box from (box, origin, boundary) on
every call (csrc/runtime.hip: make_box), exactly like the reference's C++
``get_box`` (src/box.h:208)."""
from __future__ import annotations

from typing import Iterable, Optional, Tuple, Union

import numpy as np


class Box:
    def __init__(self, box, boundary: Optional[Iterable[int]] = None, origin: Optional[Iterable[float]] = None):
        if isinstance(box, Box):  # copy constructor (box.py:107-113)
            self._box = box.box.copy()
            self._origin = box.origin.copy()
            self._boundary = box.boundary.copy()
            self._update()
            return
        self._box, self._origin = self._parse_box_origin(box, origin)
        self._update()
        self.set_boundary(boundary)

    # ---- parsing (box.py:120-223)
    @staticmethod
    def _parse_origin(origin) -> np.ndarray:
        if origin is None:
            return np.zeros(3, np.float64)
        if isinstance(origin, (list, tuple, np.ndarray)):
            o = np.array(origin, np.float64)
            if o.shape != (3,):
                raise ValueError(f"Origin must be a 3-element array, got shape {o.shape}")
            return o
        raise TypeError(f"Invalid origin type: {type(origin)}")

    @classmethod
    def _parse_box_origin(cls, box, origin) -> Tuple[np.ndarray, np.ndarray]:
        if isinstance(box, (int, float, np.integer, np.floating)):
            b = np.eye(3, dtype=np.float64) * float(box)
        elif isinstance(box, (list, tuple, np.ndarray)):
            b = np.array(box, np.float64)
            if b.shape == (3,):
                b = np.diag(b)
            elif b.shape == (3, 3):
                pass
            elif b.shape == (4, 3):  # old mdapy format: last row is the origin
                origin = np.array(b[-1])
                b = np.array(b[:-1])
            elif b.shape == (3, 4):  # ovito format: last column is the origin
                origin = np.array(b[:, -1])
                b = np.array(b[:, :-1])
            else:
                raise ValueError(f"Invalid box shape: {b.shape}")
        else:
            raise TypeError(f"Invalid box type: {type(box)}")
        return np.ascontiguousarray(b), cls._parse_origin(origin)

    def _update(self):
        b = self._box
        self._triclinic = bool(
            any(abs(b[i, j]) > 1e-10 for i in range(3) for j in range(3) if i != j) or np.any(np.diag(b) < 0)
        )  # box.py:262-276
        self._inverse = np.linalg.inv(b)
        self._volume = float(np.linalg.det(b))

    # ---- setters
    def set_box(self, box) -> None:
        if isinstance(box, (list, tuple, np.ndarray)) and np.array(box).shape not in ((3,), (3, 3)):
            raise ValueError(f"Invalid box shape: {np.array(box).shape}")
        self._box, _ = self._parse_box_origin(box, self._origin)
        self._update()

    def set_origin(self, origin) -> None:
        self._origin = self._parse_origin(origin)

    def set_boundary(self, boundary) -> None:
        if boundary is None:
            self._boundary = np.array([1, 1, 1], np.int32)
            return
        if isinstance(boundary, (list, tuple, np.ndarray)):
            p = np.array(boundary, np.int32)
            if p.shape != (3,):
                raise ValueError(f"Boundary must be a 3-element array, got shape {p.shape}")
            self._boundary = np.where(p != 0, 1, 0)  # box.py:255 (int64 result, as in the reference)
            return
        raise TypeError(f"Invalid boundary type: {type(boundary)}")

    # ---- properties
    @property
    def box(self) -> np.ndarray:
        return self._box

    @property
    def origin(self) -> np.ndarray:
        return self._origin

    @property
    def boundary(self) -> np.ndarray:
        return self._boundary

    @property
    def triclinic(self) -> bool:
        return self._triclinic

    @property
    def inverse_box(self) -> np.ndarray:
        return self._inverse

    @property
    def volume(self) -> float:
        return self._volume

    def __repr__(self) -> str:
        return (f"Box information:\n{self.box}\nOrigin: {self.origin}\nTriclinic: {self.triclinic}\n"
                f"Boundary: {self.boundary}")

    # ---- geometry
    def pbc(self, rij: np.ndarray) -> np.ndarray:
        """minimum-image convention for one displacement vector (box.py:448-467)"""
        f = np.asarray(rij, np.float64) @ self.inverse_box
        for i in range(3):
            if self.boundary[i] == 1:
                f[i] -= np.floor(f[i] + 0.5)
        return f @ self.box

    def align_to_lammps_box(self):
        """the same cell as a LAMMPS-style lower-triangular box + the rotation that maps positions into it (box.py:425-443)"""
        ax = np.linalg.norm(self.box[0])
        bx = self.box[1] @ (self.box[0] / ax)
        by = np.sqrt(np.linalg.norm(self.box[1]) ** 2 - bx ** 2)
        cx = self.box[2] @ (self.box[0] / ax)
        cy = (self.box[1] @ self.box[2] - bx * cx) / by
        cz = np.sqrt(np.linalg.norm(self.box[2]) ** 2 - cx ** 2 - cy ** 2)
        box = np.array([[ax, bx, cx], [0, by, cy], [0, 0, cz]], dtype=np.float64).T
        rotation = np.linalg.solve(self.box, box)
        return Box(box, self.boundary, self.origin), rotation

    def get_thickness(self) -> np.ndarray:
        """perpendicular thickness per axis (box.py:469-481)"""
        b = self.box
        return np.array(
            [
                self.volume / np.linalg.norm(np.cross(b[1], b[2])),
                self.volume / np.linalg.norm(np.cross(b[0], b[2])),
                self.volume / np.linalg.norm(np.cross(b[0], b[1])),
            ],
            dtype=np.float64,
        )

    def check_small_box(self, rc: float) -> np.ndarray:
        """replications needed so that every periodic thickness is >= 2 rc (box.py:483-502)"""
        t = self.get_thickness()
        repeat = np.ones(3, dtype=np.int32)
        for i in range(3):
            if self.boundary[i] == 1 and t[i] < 2 * rc:
                repeat[i] = int(np.ceil(2.0 * rc / t[i]))
        return repeat

    def is_general_box(self, tol: float = 1e-6) -> bool:
        b = self.box
        return bool(b[0, 0] <= tol or b[1, 1] <= tol or b[2, 2] <= tol or abs(b[0, 1]) > tol
                    or abs(b[0, 2]) > tol or abs(b[1, 2]) > tol)
