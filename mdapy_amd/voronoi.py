"""Voronoi tessellation — the drop-in for ``mdapy.voronoi.Voronoi`` (``get_volume``, ``get_neighbor``,
``get_cell_info``; src/mdapy/voronoi.py:20-330) and the ``Cell`` / ``Container`` views (:331-440).

The cells are built on the GPU (csrc/voronoi.hip: one wavefront per cell, faces clipped out of the sorted neighbour
bisectors).  Two conventions of the reference are kept: a triclinic box is handed over in LAMMPS alignment (a along x, b
in the xy plane) together with the rotation that takes positions there, and an open direction of such a box is widened
three-fold so that the container holds atoms that left the cell."""
import numpy as np

from . import kernels, policy
from .devarray import as_numpy
from .frame import Frame
from .parallel import get_num_threads

_NEIGHBOR_MIN_ATOMS = 50  # a smaller periodic system is replicated first (voro++ sizes its blocks from N)


def _coordinates(frame):
    return tuple(np.ascontiguousarray(as_numpy(c), dtype=np.float64) for c in policy.positions(frame))


def _aligned(box):
    """(cell matrix for the kernel, aligned box, rotation, was a rotation needed?) of a triclinic box"""
    m = box.box
    skewed = bool(np.any(np.abs(m[np.triu_indices(3, 1)]) > 1e-10) or np.any(np.diag(m) < 0))
    lammps, rotation = box.align_to_lammps_box()
    cell = lammps.box.copy()
    cell[lammps.boundary == 0] *= 3
    return cell, lammps, rotation, skewed


class Voronoi:
    def __init__(self, box, data):
        self.box, self.data = box, data

    def get_neighbor(self, a_face_area_threshold=-1.0, r_face_area_threshold=-1.0):
        """-> verlet_list, distance_list, face_area (N, W) and neighbor_number (N): atoms that share a face, optionally only
        faces above an absolute / relative area"""
        frame, box = self.data, self.box
        atoms = frame.shape[0]
        periodic = [a for a in range(3) if box.boundary[a] == 1]
        isolated = atoms < _NEIGHBOR_MIN_ATOMS and not periodic
        if isolated and atoms <= 1:
            raise AssertionError("system with all free boundary must has at least 2 atoms.")
        copies = [1, 1, 1]
        while periodic and atoms * copies[0] * copies[1] * copies[2] < _NEIGHBOR_MIN_ATOMS:
            for a in periodic:
                copies[a] += 1
        if not policy.is_single(copies):
            frame, box = policy.replica(frame, box, copies, all_columns=False)
            self._enlarge_data, self._enlarge_box = frame, box
        where = _coordinates(frame)
        cuts = (a_face_area_threshold, r_face_area_threshold, get_num_threads())
        if box.triclinic and not isolated:
            cell, lammps, rotation, skewed = _aligned(box)
            return kernels.voronoi.get_voronoi_neighbor_tri(*where, cell, lammps.origin, lammps.boundary, rotation, skewed, *cuts)
        return kernels.voronoi.get_voronoi_neighbor(*where, *policy.box_args(box), *cuts)

    def get_volume(self):
        """-> cell volume, number of faces, cavity radius (distance to the farthest vertex) per atom"""
        atoms = self.data.shape[0]
        volume, faces, radius = np.zeros(atoms), np.zeros(atoms, np.int32), np.zeros(atoms)
        where = _coordinates(self.data)
        if self.box.triclinic:
            cell, lammps, rotation, skewed = _aligned(self.box)
            kernels.voronoi.get_voronoi_volume_number_radius_tri(*where, cell, lammps.origin, lammps.boundary, rotation, volume,
                                                                 faces, radius, skewed, get_num_threads())
        else:
            kernels.voronoi.get_voronoi_volume_number_radius(*where, *policy.box_args(self.box), volume, faces, radius,
                                                             get_num_threads())
        return volume, faces, radius

    def get_cell_info(self):
        """-> per cell: faces as vertex index lists, vertex positions, volume, cavity radius, face areas"""
        if self.box.triclinic:
            raise AssertionError("Only support orthogonal box.")
        if self.data.shape[0] <= 1:
            raise AssertionError("At least has one atom.")
        return kernels.voronoi.get_cell_info(*_coordinates(self.data), *policy.box_args(self.box), get_num_threads())


class Cell:
    """One Voronoi cell: ``face_vertices`` (index lists into ``vertices``), ``vertices`` (n, 3), ``volume``,
    ``cavity_radius``, ``face_areas`` and the position ``pos`` of its atom."""

    __slots__ = ("face_vertices", "vertices", "volume", "cavity_radius", "face_areas", "pos")

    def __init__(self, face_vertices, vertices, volume, cavity_radius, face_areas, pos):
        self.face_vertices, self.vertices = face_vertices, vertices
        self.volume, self.cavity_radius = volume, cavity_radius
        self.face_areas, self.pos = face_areas, pos

    def __repr__(self):
        return f"Cell(faces={len(self.face_vertices)}, vertices={len(self.vertices)}, volume={self.volume:.6g})"


class Container:
    """The cells of all atoms of an orthogonal box, as a sequence."""

    def __init__(self, data, box):
        if isinstance(data, np.ndarray):
            assert data.ndim == 2 and data.shape[1] == 3
            data = Frame(dict(zip("xyz", data.T)))
        faces, corners, volume, radius, areas = Voronoi(box, data).get_cell_info()
        sites = np.column_stack([as_numpy(c) for c in policy.positions(data)]).astype(np.float64)
        self._data = [Cell(faces[i], np.array(corners[i], np.float64).reshape(-1, 3), volume[i], radius[i],
                           np.array(areas[i], np.float64), sites[i]) for i in range(len(sites))]

    def __getitem__(self, index):
        return self._data[index]

    def __len__(self):
        return len(self._data)

    def __iter__(self):
        return iter(self._data)
