"""Polycrystal builder.  Mirrors ``mdapy.create_polycrystal.CreatePolycrystal`` (src/mdapy/create_polycrystal.py:20-850)
for metallic grains: Voronoi tessellation of the seeds, every cell filled with the rotated unit cell, overlapping
atoms at the grain boundaries removed, atoms wrapped into the box.

Where the work is done
* Voronoi cells of the seeds (faces, vertices, volumes, cavity radii): ``_voronoi.get_cell_info`` (HIP);
* filling a grain: ``_polycrystal.transform_and_filter`` (HIP) — rotate, translate, half-space test, compaction;
* grain-boundary overlaps: ``_neighbor.filter_overlap_atom`` (HIP); wrapping: ``System.wrap_pos`` (HIP).

The face planes of a grain come from the vertices of its Voronoi cell exactly as in the reference
(``_get_plane_equation_coeffs_for_cell``, :207-256): normal = cross product of two edges at the first vertex of the face,
oriented so that the seed is on the negative side.  The cells are those of ``mdapy_amd.voronoi.Container`` (HIP); their
vertices agree with voro++'s to rounding, so only atoms closer than that to a grain-boundary plane can be assigned
differently.

Graphene-decorated grain boundaries (``add_graphene=True``, :331-520, ``filter_overlap_atom_with_grain``) are refused.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple, Union

import numpy as np

from . import _neighbor, _polycrystal
from . import tool_function as tool
from .box import Box
from .frame import Frame
from .parallel import get_num_threads
from .voronoi import Container


class CreatePolycrystal:
    def __init__(self, unitcell, box: Union[int, float, Iterable[float], np.ndarray, Box], seed_number: int,
                 seed_position: Optional[np.ndarray] = None, theta_list: Optional[np.ndarray] = None,
                 randomseed: Optional[int] = None, metal_overlap_dis: Optional[float] = None, add_graphene: bool = False,
                 metal_gra_overlap_dis: float = 3.0, face_threshold: float = 0.0, need_rotation: bool = True) -> None:
        self.unitcell = unitcell
        self.box = Box(box)
        if sum(self.box.boundary) != 3:
            raise ValueError("Free boundary condition is not supported.")
        if self.box.triclinic:
            raise ValueError("Triclinic box is not supported")
        if add_graphene:
            raise NotImplementedError("add_graphene=True (graphene at the grain boundaries) is not available in mdapy_amd")
        self.seed_number = int(seed_number)
        self.metal_overlap_dis = metal_overlap_dis
        self.add_graphene = False
        self.metal_gra_overlap_dis = metal_gra_overlap_dis
        self.need_rotation = need_rotation
        self.face_threshold = face_threshold
        if randomseed is None:
            randomseed = np.random.randint(0, 1_000_000_000)
        self.randomseed = int(randomseed)
        self.rng = np.random.default_rng(self.randomseed)
        if seed_position is None:  # same draws, same order as the reference (:126-147)
            self.seed_position = self.rng.random((self.seed_number, 3)) * np.diag(self.box.box)
        else:
            seed_position = np.asarray(seed_position, float)
            if seed_position.shape != (self.seed_number, 3):
                raise ValueError(f"seed_position shape must be ({self.seed_number}, 3), got {seed_position.shape}")
            self.seed_position = seed_position
        if theta_list is None:
            self.theta_list = self.rng.uniform(-180, 180, (self.seed_number, 3))
        else:
            theta_list = np.asarray(theta_list, float)
            if theta_list.shape != (self.seed_number, 3):
                raise ValueError(f"theta_list shape must be ({self.seed_number}, 3), got {theta_list.shape}")
            self.theta_list = theta_list

    @staticmethod
    def _get_rotation_matrix(theta_deg: float, axis_tuple: Tuple[float, float, float]) -> np.ndarray:
        """Rodrigues' formula (create_polycrystal.py:152-205)"""
        theta = np.radians(theta_deg)
        axis = np.array(axis_tuple, dtype=float)
        norm = np.linalg.norm(axis)
        if norm == 0:
            raise ValueError("Rotation axis must be non-zero")
        x, y, z = axis / norm
        c, s = np.cos(theta), np.sin(theta)
        C = 1 - c
        return np.array([[c + C * x * x, C * x * y - s * z, C * x * z + s * y],
                         [C * y * x + s * z, c + C * y * y, C * y * z - s * x],
                         [C * z * x - s * y, C * z * y + s * x, c + C * z * z]], dtype=float)

    @staticmethod
    def _get_plane_equation_coeffs_for_cell(cell) -> np.ndarray:
        """(n_faces, 4) rows (a, b, c, d) with a*x + b*y + c*z + d < 0 inside the cell (create_polycrystal.py:207-256)"""
        coeffs = np.zeros((len(cell.face_vertices), 4))
        for i, face in enumerate(cell.face_vertices):
            p1, p2, p3 = cell.vertices[face[0]], cell.vertices[face[1]], cell.vertices[face[2]]
            n = np.cross(p2 - p1, p3 - p1)
            norm_n = np.linalg.norm(n)
            if norm_n < 1e-10:
                raise ValueError(f"Degenerate face vertices at face {i}")
            n = n / norm_n
            d = -np.dot(n, p1)
            if np.dot(n, cell.pos) + d > 0:  # normal points outward: the seed is on the negative side
                n, d = -n, -d
            coeffs[i, :3] = n
            coeffs[i, 3] = d
        return coeffs

    def _get_pos(self):
        r_max = max(cell.cavity_radius for cell in self.con)
        thickness = self.unitcell.box.get_thickness()
        replicate_nums = np.ceil(r_max / thickness).astype(int)  # :597-600
        data, _ = tool._replicate_pos(self.unitcell.data, self.unitcell.box, *replicate_nums)
        x, y, z = (np.ascontiguousarray(data[c].to_numpy(), dtype=np.float64) for c in ("x", "y", "z"))
        pos_center = np.array([x.mean(), y.mean(), z.mean()])
        pos_list, grain_list = [], []
        for n in range(self.seed_number):
            if self.need_rotation:  # :289-301
                rot = (self._get_rotation_matrix(self.theta_list[n, 0], (1.0, 0.0, 0.0))
                       @ self._get_rotation_matrix(self.theta_list[n, 1], (0.0, 1.0, 0.0))
                       @ self._get_rotation_matrix(self.theta_list[n, 2], (0.0, 0.0, 1.0)))
            else:
                rot = self._get_rotation_matrix(0, (1.0, 0.0, 0.0))
            cell = self.con[n]
            pos = _polycrystal.transform_and_filter(x, y, z, rot, pos_center, cell.pos,
                                                    self._get_plane_equation_coeffs_for_cell(cell), get_num_threads())
            pos_list.append(pos)
            grain_list.append(np.full(len(pos), n + 1, np.int32))
        return np.vstack(pos_list), np.concatenate(grain_list)

    def compute(self, verbose: bool = False):
        """-> System with columns element (when the unit cell has one), x, y, z, grain_id, type  (:684-848)"""
        from .system import System

        origin = self.box.origin.copy()
        self.con = Container(np.ascontiguousarray(self.seed_position, dtype=np.float64), Box(self.box.box))
        self.volume = np.array([cell.volume for cell in self.con])
        if verbose:
            print(f"  Number of grains: {self.seed_number}\n  Average volume:   {self.volume.mean():>10.2f} A^3")
        pos, grain_id = self._get_pos()
        n_generated = len(pos)
        x, y, z = pos[:, 0] + origin[0], pos[:, 1] + origin[1], pos[:, 2] + origin[2]
        type_list = np.ones(n_generated, np.int32)
        if self.metal_overlap_dis is not None:  # :803-815
            keep = np.asarray(_neighbor.filter_overlap_atom(np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(z),
                                                            self.box.box, self.box.origin, self.box.boundary,
                                                            float(self.metal_overlap_dis), get_num_threads()), bool)
            x, y, z, grain_id, type_list = x[keep], y[keep], z[keep], grain_id[keep], type_list[keep]
        if verbose:
            print(f"  Total atoms generated: {n_generated:,}; removed: {n_generated - len(x):,}")
        cols = {"x": x, "y": y, "z": z, "grain_id": grain_id, "type": type_list}
        if "element" in self.unitcell.data.columns:
            element = self.unitcell.data["element"].to_numpy()[0]
            cols = {"element": np.full(len(x), element), **cols}
        system = System(data=Frame(cols), box=self.box)
        system.wrap_pos()
        return system
