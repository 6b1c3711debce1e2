"""Simulation box, host side — the drop-in for ``mdapy.box.Box`` (src/mdapy/box.py:93-502).

Rows of the 3x3 matrix are the cell vectors a, b, c; ``origin`` is the corner they start from; ``boundary`` holds 1 for a
periodic direction and 0 for an open one.  The kernels rebuild their own device-side box from these three arrays on every
call (csrc/runtime.hip: make_box), as the reference's C++ does (src/box.h:208)."""
import numpy as np

_SCALARS = (int, float, np.integer, np.floating)
_SEQUENCES = (list, tuple, np.ndarray)


def _three(values, what, kind):
    """a 3-vector of the given dtype from a list / tuple / array, with the reference's error texts"""
    if not isinstance(values, _SEQUENCES):
        raise TypeError(f"Invalid {what.lower()} type: {type(values)}")
    vec = np.array(values, kind)
    if vec.shape != (3,):
        raise ValueError(f"{what} must be a 3-element array, got shape {vec.shape}")
    return vec


def _cell_and_origin(spec, origin):
    """(3x3 cell, origin-or-None) from any of the accepted box descriptions: one edge length, three edge lengths, a 3x3
    matrix, a 4x3 matrix whose last row is the origin (old mdapy files), a 3x4 matrix whose last column is (OVITO)"""
    if isinstance(spec, _SCALARS):
        return np.eye(3, dtype=np.float64) * float(spec), origin
    if not isinstance(spec, _SEQUENCES):
        raise TypeError(f"Invalid box type: {type(spec)}")
    m = np.array(spec, np.float64)
    if m.shape == (3,):
        return np.diag(m), origin
    if m.shape == (3, 3):
        return m, origin
    if m.shape == (4, 3):
        return np.array(m[:3]), np.array(m[3])
    if m.shape == (3, 4):
        return np.array(m[:, :3]), np.array(m[:, 3])
    raise ValueError(f"Invalid box shape: {m.shape}")


class Box:
    def __init__(self, box, boundary=None, origin=None):
        if isinstance(box, Box):  # copy
            self._box, self._origin, self._boundary = box.box.copy(), box.origin.copy(), box.boundary.copy()
            self._derive()
            return
        cell, start = _cell_and_origin(box, origin)
        self._box = np.ascontiguousarray(cell)
        self._origin = np.zeros(3, np.float64) if start is None else _three(start, "Origin", np.float64)
        self._derive()
        self.set_boundary(boundary)

    def _derive(self):
        cell = self._box
        off_diagonal = cell[~np.eye(3, dtype=bool)]
        # sheared, or mirrored along an axis: the general (triclinic) code path of the kernels (src/box.h:216-222)
        self._triclinic = bool(np.any(np.abs(off_diagonal) > 1e-10) or np.any(np.diag(cell) < 0))
        self._inverse = np.linalg.inv(cell)
        self._volume = float(np.linalg.det(cell))
        self._thickness = None  # (made when first asked for, with the inverse and the volume a function of the cell alone)

    # ---- setters
    def set_box(self, box):
        if isinstance(box, _SEQUENCES) and np.array(box).shape not in ((3,), (3, 3)):
            raise ValueError(f"Invalid box shape: {np.array(box).shape}")
        self._box = np.ascontiguousarray(_cell_and_origin(box, None)[0])
        self._derive()

    def set_origin(self, origin):
        self._origin = np.zeros(3, np.float64) if origin is None else _three(origin, "Origin", np.float64)

    def set_boundary(self, boundary):
        if boundary is None:
            self._boundary = np.array([1, 1, 1], np.int32)
        else:
            flags = _three(boundary, "Boundary", np.int32)
            self._boundary = np.where(flags != 0, 1, 0)  # (platform integer, like the reference's)

    # ---- read access
    box = property(lambda self: self._box)
    origin = property(lambda self: self._origin)
    boundary = property(lambda self: self._boundary)
    triclinic = property(lambda self: self._triclinic)
    inverse_box = property(lambda self: self._inverse)
    volume = property(lambda self: self._volume)

    def __repr__(self):
        return (f"Box information:\n{self.box}\nOrigin: {self.origin}\nTriclinic: {self.triclinic}\n"
                f"Boundary: {self.boundary}")

    # ---- geometry
    def pbc(self, rij):
        """minimum image of one displacement vector"""
        frac = np.asarray(rij, np.float64) @ self.inverse_box
        periodic = self.boundary == 1
        frac[periodic] -= np.floor(frac[periodic] + 0.5)
        return frac @ self.box

    def align_to_lammps_box(self):
        """(the same cell with a along x and b in the xy plane, the rotation that takes positions into it)"""
        a, b, c = self.box
        ax = np.linalg.norm(a)
        ahat = a / ax
        bx = b @ ahat
        by = np.sqrt(np.linalg.norm(b) ** 2 - bx ** 2)
        cx = c @ ahat
        cy = (b @ c - bx * cx) / by
        cz = np.sqrt(np.linalg.norm(c) ** 2 - cx ** 2 - cy ** 2)
        lower = np.array([[ax, bx, cx], [0, by, cy], [0, 0, cz]], dtype=np.float64).T
        return Box(lower, self.boundary, self.origin), np.linalg.solve(self.box, lower)

    def get_thickness(self):
        """distance between opposite faces, per direction: volume over the area of the face the other two vectors span"""
        # (every analysis asks two or three times per call — the 15 A and 2 rc replication rules — and three numpy cross products
        # were 0.2 ms of a 1.8 ms frame through System: kept with the cell they belong to)
        if self._thickness is None:
            a, b, c = self.box
            faces = (np.cross(b, c), np.cross(a, c), np.cross(a, b))
            self._thickness = np.array([self.volume / np.linalg.norm(f) for f in faces], dtype=np.float64)
        return self._thickness.copy()

    def check_small_box(self, rc):
        """copies per axis after which every periodic direction is at least two cutoffs thick"""
        from .policy import axis_copies

        return axis_copies(self, 2 * rc)

    def is_general_box(self, tol=1e-6):
        m = self.box
        upper = (m[0, 1], m[0, 2], m[1, 2])
        return bool(min(m[0, 0], m[1, 1], m[2, 2]) <= tol or max(abs(v) for v in upper) > tol)
