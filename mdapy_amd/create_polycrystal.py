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

Graphene-decorated grain boundaries (``add_graphene=True``, :331-560): a honeycomb sheet is laid on every cell face larger
than ``face_threshold`` (rotated onto the face normal, centred on the face, cut to the face polygon by 2-D ray casting), and
the metal / carbon overlaps are resolved by ``_neighbor.filter_overlap_atom_with_grain`` (HIP; the reference's sweep in its
serial order).
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
        self.seed_number = int(seed_number)
        self.metal_overlap_dis = metal_overlap_dis
        self.add_graphene = bool(add_graphene)
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

    @classmethod
    def _rotation_onto(cls, source: np.ndarray, target: np.ndarray) -> np.ndarray:
        """rotation that turns the unit vector ``source`` into ``target`` (create_polycrystal.py:394-446)"""
        v1, v2 = source / np.linalg.norm(source), target / np.linalg.norm(target)
        dot = float(np.dot(v1, v2))
        if np.isclose(dot, 1.0, atol=1e-6):
            return np.eye(3)
        if np.isclose(dot, -1.0, atol=1e-6):  # half a turn about any axis perpendicular to v1
            axis = np.cross(v1, np.array([1.0, 0.0, 0.0]) if abs(v1[0]) < 0.9 else np.array([0.0, 1.0, 0.0]))
            return cls._get_rotation_matrix(180.0, tuple(axis / np.linalg.norm(axis)))
        axis = np.cross(v1, v2)
        axis = axis / np.linalg.norm(axis)
        return cls._get_rotation_matrix(float(np.degrees(np.arccos(np.clip(dot, -1.0, 1.0)))), tuple(axis))

    @staticmethod
    def _points_in_polygon_2d(polygon: np.ndarray, points: np.ndarray) -> np.ndarray:
        """even-odd ray casting in single precision, points on a vertex count as inside (create_polycrystal.py:522-580)"""
        polygon = np.asarray(polygon, dtype=np.float32)
        points = np.asarray(points, dtype=np.float32)
        v1 = polygon[None, :, :]
        v2 = np.roll(polygon, -1, axis=0)[None, :, :]
        pts = points[:, None, :]
        on_vertex = np.any(np.all(np.isclose(pts, v1, atol=1e-6), axis=2), axis=1)
        crosses = (v1[:, :, 1] > pts[:, :, 1]) != (v2[:, :, 1] > pts[:, :, 1])
        x_at = (v2[:, :, 0] - v1[:, :, 0]) * (pts[:, :, 1] - v1[:, :, 1]) / (v2[:, :, 1] - v1[:, :, 1] + 1e-10) + v1[:, :, 0]
        hits = np.sum(crosses & (pts[:, :, 0] < x_at), axis=1)
        return (hits % 2 == 1) | on_vertex

    def _filter_atoms_in_polygon(self, points: np.ndarray, polygon_vertices: np.ndarray, face_normal: np.ndarray) -> np.ndarray:
        """points within 0.5 A of the face plane whose projection lies in the face polygon (create_polycrystal.py:448-520)"""
        ez = face_normal / np.linalg.norm(face_normal)
        centre = polygon_vertices.mean(axis=0)
        ex = polygon_vertices[0] - centre
        ex = ex - np.dot(ex, ez) * ez
        if np.linalg.norm(ex) < 1e-8:
            ex = polygon_vertices[1] - centre
            ex = ex - np.dot(ex, ez) * ez
        ex = ex / np.linalg.norm(ex)
        frame = np.array([ex, np.cross(ez, ex), ez])
        poly_local = (polygon_vertices - centre) @ frame.T
        pts_local = (points - centre) @ frame.T
        keep = (np.abs(pts_local[:, 2]) < 0.5) & self._points_in_polygon_2d(poly_local[:, :2], pts_local[:, :2])
        return points[keep]

    def _generate_gra_atoms(self, cell, gra_pos: np.ndarray, coeffs: np.ndarray) -> np.ndarray:
        """carbon atoms on the faces of one cell (create_polycrystal.py:331-392)"""
        sheets = []
        for f in range(coeffs.shape[0]):
            if cell.face_areas[f] <= self.face_threshold:
                continue
            verts = cell.vertices[cell.face_vertices[f]]
            normal = coeffs[f, :3] / np.linalg.norm(coeffs[f, :3])
            rot = self._rotation_onto(np.array([0.0, 0.0, 1.0]), normal)
            sheet = gra_pos @ rot.T
            sheet = sheet - sheet.mean(axis=0) + verts.mean(axis=0)
            inside = self._filter_atoms_in_polygon(sheet, verts, normal)
            if len(inside):
                sheets.append(inside)
        assert len(sheets) > 0, "No graphene atoms generated"
        return np.vstack(sheets)

    def _get_pos(self):
        r_max = max(cell.cavity_radius for cell in self.con)
        thickness = self.unitcell.box.get_thickness()
        replicate_nums = np.ceil(r_max / thickness).astype(int)  # :597-600
        data, _ = tool._replicate_pos(self.unitcell.data, self.unitcell.box, *replicate_nums)
        x, y, z = (np.ascontiguousarray(data[c].to_numpy(), dtype=np.float64) for c in ("x", "y", "z"))
        pos_center = np.array([x.mean(), y.mean(), z.mean()])
        gra_pos = None
        if self.add_graphene:  # a sheet whose (x, y) extent covers 2 r_max: enough for any face once centred on it (:606-621)
            from .build_lattice import lattice_positions

            gra_lattice = 1.42 * 3 ** 0.5  # hexagonal in-plane parameter for a 1.42 A C-C bond
            target = 2.0 * r_max
            gra_pos, _ = lattice_positions("graphene", gra_lattice, int(np.ceil(target / gra_lattice)),
                                           int(np.ceil(target / (gra_lattice * 3 ** 0.5 / 2.0))), 1, c=1.0)
        pos_list, grain_list, type_list = [], [], []
        for n in range(self.seed_number):
            if self.need_rotation:  # :289-301
                rot = (self._get_rotation_matrix(self.theta_list[n, 0], (1.0, 0.0, 0.0))
                       @ self._get_rotation_matrix(self.theta_list[n, 1], (0.0, 1.0, 0.0))
                       @ self._get_rotation_matrix(self.theta_list[n, 2], (0.0, 0.0, 1.0)))
            else:
                rot = self._get_rotation_matrix(0, (1.0, 0.0, 0.0))
            cell = self.con[n]
            coeffs = self._get_plane_equation_coeffs_for_cell(cell)
            pos = _polycrystal.transform_and_filter(x, y, z, rot, pos_center, cell.pos, coeffs, get_num_threads())
            pos_list.append(pos)
            type_list.append(np.ones(len(pos), np.int32))
            n_grain = len(pos)
            if self.add_graphene:
                carbon = self._generate_gra_atoms(cell, gra_pos, coeffs)
                pos_list.append(carbon)
                type_list.append(np.full(len(carbon), 2, np.int32))
                n_grain += len(carbon)
            grain_list.append(np.full(n_grain, n + 1, np.int32))
        return np.vstack(pos_list), np.concatenate(grain_list), np.concatenate(type_list)

    def compute(self, verbose: bool = False):
        """-> System with columns element (when the unit cell has one), x, y, z, grain_id, type  (:684-848)"""
        from .system import System

        origin = self.box.origin.copy()
        self.con = Container(np.ascontiguousarray(self.seed_position, dtype=np.float64), Box(self.box.box))
        self.volume = np.array([cell.volume for cell in self.con])
        if verbose:
            print(f"  Number of grains: {self.seed_number}\n  Average volume:   {self.volume.mean():>10.2f} A^3")
        pos, grain_id, type_list = self._get_pos()
        n_generated = len(pos)
        x, y, z = pos[:, 0] + origin[0], pos[:, 1] + origin[1], pos[:, 2] + origin[2]
        if self.add_graphene:  # :770-801: metal-metal (default 2.0 A), C-C 1.4 A, metal-C
            mm = float(self.metal_overlap_dis) if self.metal_overlap_dis is not None else 2.0
            keep = np.asarray(_neighbor.filter_overlap_atom_with_grain(np.ascontiguousarray(x), np.ascontiguousarray(y),
                                                                       np.ascontiguousarray(z), type_list, grain_id, self.box.box,
                                                                       self.box.origin, self.box.boundary, mm, 1.4,
                                                                       float(self.metal_gra_overlap_dis), get_num_threads()), bool)
            x, y, z, grain_id, type_list = x[keep], y[keep], z[keep], grain_id[keep], type_list[keep]
        elif self.metal_overlap_dis is not None:  # :803-815
            keep = np.asarray(_neighbor.filter_overlap_atom(np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(z),
                                                            self.box.box, self.box.origin, self.box.boundary,
                                                            float(self.metal_overlap_dis), get_num_threads()), bool)
            x, y, z, grain_id, type_list = x[keep], y[keep], z[keep], grain_id[keep], type_list[keep]
        if verbose:
            print(f"  Total atoms generated: {n_generated:,}; removed: {n_generated - len(x):,}")
        cols = {"x": x, "y": y, "z": z, "grain_id": grain_id, "type": type_list}
        if "element" in self.unitcell.data.columns:
            element = self.unitcell.data["element"].to_numpy()[0]
            cols = {"element": np.where(type_list == 2, "C", element), **cols}  # type 1 = the unit cell's element, 2 = carbon
        system = System(data=Frame(cols), box=self.box)
        system.wrap_pos()
        return system
