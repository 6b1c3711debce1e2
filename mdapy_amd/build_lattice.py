"""Crystal builder for the benchmark / test inputs of the hot path.

Covers the standard-orientation subset of ``mdapy.build_crystal``
(src/mdapy/build_lattice.py:657-907) that the neighbor/structure-analysis tests
use: sc, fcc, bcc, diamond (cubic cell) and hcp (2-atom hexagonal cell).  Atom
order is cell-major over (ix, iy, iz) with the basis innermost
(src/repeat_cell.cpp:41-59); positions are ``basis @ cell + shift`` evaluated
exactly as in the reference so that knife-edge cutoffs give identical counts.
Miller-index orientations and multi-species ordered structures are out of scope
(SURVEY.md §2.1 "Builders")."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _repeat_cell
from .box import Box
from .devarray import have_gpu
from .frame import Frame
from .parallel import get_num_threads

_SQRT3 = np.sqrt(3.0)

_CUBIC = {
    "sc": [[0.0, 0.0, 0.0]],
    "fcc": [[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]],  # build_lattice.py:38-52
    "bcc": [[0.0, 0.0, 0.0], [0.5, 0.5, 0.5]],
    "diamond": [[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5],
                [0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]],
}


def unit_cell(structure: str, a: float, c: Optional[float] = None):
    s = structure.lower()
    if s in _CUBIC:
        return a * np.eye(3), np.array(_CUBIC[s], dtype=np.float64)
    if s == "hcp":  # build_lattice.py:194-220, c/a = sqrt(8/3) by default (:290)
        if c is None:
            c = a * float(np.sqrt(8 / 3))
        box = np.array([[a, 0.0, 0.0], [-0.5 * a, 0.5 * _SQRT3 * a, 0.0], [0.0, 0.0, c]])
        return box, np.array([[0.0, 0.0, 0.0], [1.0 / 3.0, 2.0 / 3.0, 0.5]])
    if s == "graphene":  # build_lattice.py:238-251: one honeycomb layer in the hexagonal cell, c = vacuum spacing (must be given)
        if c is None:
            raise ValueError("graphene needs c (the spacing of the periodic images along z)")
        box = np.array([[a, 0.0, 0.0], [-0.5 * a, 0.5 * _SQRT3 * a, 0.0], [0.0, 0.0, c]])
        return box, np.array([[0.0, 0.0, 0.0], [1.0 / 3.0, 2.0 / 3.0, 0.0]])
    raise ValueError(f"Unrecognized structure '{structure}'. Supported here: {sorted(_CUBIC) + ['hcp', 'graphene']}")


def lattice_positions(structure: str, a: float, nx: int = 1, ny: int = 1, nz: int = 1, c: Optional[float] = None):
    """(positions (N,3), box (3,3)) of an nx x ny x nz supercell — numpy only (no GPU needed)."""
    cell, basis = unit_cell(structure, a, c)
    old_pos = basis @ cell  # build_lattice.py:887
    sx = np.arange(nx)[:, None, None, None] * cell[0] + np.arange(ny)[None, :, None, None] * cell[1] \
        + np.arange(nz)[None, None, :, None] * cell[2]  # ((ix*a1 + iy*a2) + iz*a3), repeat_cell.cpp:48-50
    pos = (old_pos[None, None, None, :, :] + sx[:, :, :, None, :]).reshape(-1, 3)
    return np.ascontiguousarray(pos), cell * np.array([nx, ny, nz]).reshape(3, 1)


def build_crystal(name, structure: str, a: float, nx: int = 1, ny: int = 1, nz: int = 1, c: Optional[float] = None):
    """Build a ``System`` holding an nx x ny x nz supercell (standard orientation)."""
    from .system import System

    if not isinstance(name, str):
        raise TypeError("only single-element crystals are supported here; pass one element symbol")
    cell, basis = unit_cell(structure, a, c)
    old_pos = np.ascontiguousarray(basis @ cell)
    if have_gpu():
        new_pos = np.zeros(old_pos.shape[0] * nx * ny * nz * 3, dtype=np.float64)
        _repeat_cell.repeat_cell(new_pos, cell, old_pos, nx, ny, nz, get_num_threads())
        new_pos = new_pos.reshape((-1, 3))
    else:  # pure numpy evaluation of the same expression (host-logic tests without a GPU)
        new_pos, _ = lattice_positions(structure, a, nx, ny, nz, c)
    new_box = cell * np.array([nx, ny, nz]).reshape((3, 1))
    elements = np.full(new_pos.shape[0], name, dtype=object)
    data = Frame({"x": new_pos[:, 0], "y": new_pos[:, 1], "z": new_pos[:, 2], "element": elements})
    return System(data=data, box=Box(new_box))
