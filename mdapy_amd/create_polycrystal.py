"""Polycrystal builder.  Mirrors ``mdapy.create_polycrystal.CreatePolycrystal`` (src/mdapy/create_polycrystal.py:20-850)
for metallic grains: Voronoi tessellation of the seeds, every cell filled with the rotated unit cell, overlapping
atoms at the grain boundaries removed, atoms wrapped into the box.

Where the work is done
* cell volumes and cavity radii of the seeds: ``_voronoi.get_voronoi_volume_number_radius`` (HIP);
* filling a grain: ``_polycrystal.transform_and_filter`` (HIP) — rotate, translate, half-space test, compaction;
* grain-boundary overlaps: ``_neighbor.filter_overlap_atom`` (HIP); wrapping: ``System.wrap_pos`` (HIP).

The face planes of a grain are written down directly as the bisector planes between its seed and every periodic image of
a seed within twice its cavity radius (every face of the cell lies on one of them; the others are redundant half-spaces).
The reference derives the same planes from the first three vertices of each face of voro++'s cell
(``_get_plane_equation_coeffs_for_cell``, :207-256), which agrees to rounding; atoms closer than that to a grain boundary
plane are the only ones that can be assigned differently.

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
from .voronoi import Voronoi

MAX_PLANES = 1024  # capacity of the device kernel's plane table


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

    def _cell_planes(self, i: int, radius: float) -> np.ndarray:
        """(n, 4) rows (a, b, c, d), a*x + b*y + c*z + d < 0 inside the Voronoi cell of seed ``i`` (box frame, origin at 0)"""
        L = np.diag(self.box.box)
        p = self.seed_position[i]
        reach = 2.0 * radius * (1.0 + 1e-9) + 1e-9
        span = [np.arange(-int(np.ceil(reach / L[a])) - 1, int(np.ceil(reach / L[a])) + 2) for a in range(3)]
        shifts = np.stack(np.meshgrid(*span, indexing="ij"), axis=-1).reshape(-1, 3) * L
        others = (self.seed_position[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
        d = others - p
        r = np.linalg.norm(d, axis=1)
        keep = (r > 0) & (r <= reach)
        d, r, others = d[keep], r[keep], others[keep]
        order = np.argsort(r, kind="stable")  # nearest planes first: most atoms are rejected by the first few tests
        d, r, others = d[order], r[order], others[order]
        if len(r) > MAX_PLANES:
            raise ValueError(f"grain {i}: {len(r)} seed images within twice the cavity radius (more than {MAX_PLANES})")
        u = d / r[:, None]
        mid = p + 0.5 * d
        return np.ascontiguousarray(np.c_[u, -(u * mid).sum(1)])

    def _get_pos(self, radius: np.ndarray):
        r_max = float(radius.max())
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
            pos = _polycrystal.transform_and_filter(x, y, z, rot, pos_center, self.seed_position[n],
                                                    self._cell_planes(n, float(radius[n])), get_num_threads())
            pos_list.append(pos)
            grain_list.append(np.full(len(pos), n + 1, np.int32))
        return np.vstack(pos_list), np.concatenate(grain_list)

    def compute(self, verbose: bool = False):
        """-> System with columns element (when the unit cell has one), x, y, z, grain_id, type  (:684-848)"""
        from .system import System

        origin = self.box.origin.copy()
        seeds = Frame({"x": self.seed_position[:, 0], "y": self.seed_position[:, 1], "z": self.seed_position[:, 2]})
        self.volume, _, self.cavity_radius = Voronoi(Box(self.box.box), seeds).get_volume()
        if verbose:
            print(f"  Number of grains: {self.seed_number}\n  Average volume:   {self.volume.mean():>10.2f} A^3")
        pos, grain_id = self._get_pos(np.asarray(self.cavity_radius))
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
