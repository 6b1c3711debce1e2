"""Voronoi cell volume / face count / cavity radius.  Mirrors ``mdapy.voronoi.Voronoi.get_volume``
``get_neighbor`` and ``get_cell_info`` (src/mdapy/voronoi.py:20-330) and the ``Cell`` / ``Container`` views (:331-440)."""
from __future__ import annotations

import numpy as np

from . import _voronoi
from . import tool_function as tool
from .box import Box
from .devarray import as_numpy
from .frame import Frame
from .parallel import get_num_threads


class Voronoi:
    def __init__(self, box: Box, data: Frame):
        self.box = box
        self.data = data

    def get_neighbor(self, a_face_area_threshold: float = -1.0, r_face_area_threshold: float = -1.0):
        """-> verlet_list, distance_list, face_area (N, W), neighbor_number (N) (voronoi.py:20-110)"""
        num_t = get_num_threads()
        repeat = [1, 1, 1]
        N = self.data.shape[0]
        nopbc = False
        if N < 50:
            if sum(self.box.boundary) > 0:
                while np.prod(repeat) * N < 50:
                    for i in range(3):
                        if self.box.boundary[i] == 1:
                            repeat[i] += 1
            else:
                assert N > 1, "system with all free boundary must has at least 2 atoms."
                nopbc = True
        data, box = self.data, self.box
        if sum(repeat) != 3:
            self._enlarge_data, self._enlarge_box = tool._replicate_pos(data, box, *repeat)
            data, box = self._enlarge_data, self._enlarge_box
        x, y, z = (np.ascontiguousarray(as_numpy(a), dtype=np.float64) for a in tool.xyz(data))
        if box.triclinic and not nopbc:
            b = box.box
            need_rotation = bool(abs(b[0, 1]) > 1e-10 or abs(b[0, 2]) > 1e-10 or abs(b[1, 2]) > 1e-10 or b[0, 0] < 0
                                 or b[1, 1] < 0 or b[2, 2] < 0)
            lbox, rotation = box.align_to_lammps_box()
            bb = lbox.box.copy()
            for i in range(3):
                if lbox.boundary[i] == 0:
                    bb[i] *= 3
            return _voronoi.get_voronoi_neighbor_tri(x, y, z, bb, lbox.origin, lbox.boundary, rotation, need_rotation,
                                                     a_face_area_threshold, r_face_area_threshold, num_t)
        return _voronoi.get_voronoi_neighbor(x, y, z, box.box, box.origin, box.boundary, a_face_area_threshold,
                                             r_face_area_threshold, num_t)

    def get_volume(self):
        n = self.data.shape[0]
        volume = np.zeros(n)
        neighbor_number = np.zeros(n, np.int32)
        cavity_radius = np.zeros(n)
        x, y, z = (np.ascontiguousarray(as_numpy(a), dtype=np.float64) for a in tool.xyz(self.data))
        if self.box.triclinic:
            b = self.box.box
            need_rotation = bool(abs(b[0, 1]) > 1e-10 or abs(b[0, 2]) > 1e-10 or abs(b[1, 2]) > 1e-10 or b[0, 0] < 0
                                 or b[1, 1] < 0 or b[2, 2] < 0)
            box, rotation = self.box.align_to_lammps_box()
            bb = box.box.copy()
            for i in range(3):
                if box.boundary[i] == 0:
                    bb[i] *= 3
            _voronoi.get_voronoi_volume_number_radius_tri(x, y, z, bb, box.origin, box.boundary, rotation, volume,
                                                          neighbor_number, cavity_radius, need_rotation, get_num_threads())
        else:
            _voronoi.get_voronoi_volume_number_radius(x, y, z, self.box.box, self.box.origin, self.box.boundary, volume,
                                                      neighbor_number, cavity_radius, get_num_threads())
        return volume, neighbor_number, cavity_radius

    def get_cell_info(self):
        """-> face_vertices_indices, face_vertices_positions, volume, radius, face_areas (voronoi.py:184-246)"""
        assert not self.box.triclinic, "Only support orthogonal box."
        assert self.data.shape[0] > 1, "At least has one atom."
        x, y, z = (np.ascontiguousarray(as_numpy(a), dtype=np.float64) for a in tool.xyz(self.data))
        return _voronoi.get_cell_info(x, y, z, self.box.box, self.box.origin, self.box.boundary, get_num_threads())


class Cell:
    """One Voronoi cell (voronoi.py:331-369): faces as index lists into ``vertices``, volume, cavity radius, face areas,
    position of the atom."""

    def __init__(self, face_vertices, vertices, volume, cavity_radius, face_areas, pos):
        self.face_vertices = face_vertices
        self.vertices = vertices
        self.volume = volume
        self.cavity_radius = cavity_radius
        self.face_areas = face_areas
        self.pos = pos

    def __repr__(self):
        return f"Cell(faces={len(self.face_vertices)}, vertices={len(self.vertices)}, volume={self.volume:.6g})"


class Container:
    """The cells of every atom, list-like (voronoi.py:372-440)."""

    def __init__(self, data, box: Box):
        if isinstance(data, np.ndarray):
            assert data.ndim == 2 and data.shape[1] == 3
            data = Frame({"x": data[:, 0], "y": data[:, 1], "z": data[:, 2]})
        fvi, fvp, volume, radius, face_areas = Voronoi(box, data).get_cell_info()
        x, y, z = (as_numpy(a) for a in tool.xyz(data))
        self._data = [Cell(fvi[i], np.array(fvp[i], np.float64).reshape(-1, 3), volume[i], radius[i], np.array(face_areas[i], np.float64),
                           np.array([x[i], y[i], z[i]], np.float64)) for i in range(data.shape[0])]

    def __getitem__(self, index: int):
        return self._data[index]

    def __len__(self):
        return len(self._data)

    def __iter__(self):
        return iter(self._data)

