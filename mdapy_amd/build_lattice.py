"""Crystal builder for the benchmark and test inputs of the hot path — the standard-orientation subset of
``mdapy.build_crystal`` (src/mdapy/build_lattice.py:657-907): sc, fcc, bcc, diamond in the cubic cell, hcp and a graphene
layer in the hexagonal cell.  Atoms come cell by cell, (ix, iy, iz)-lexicographic with the basis innermost, at
``basis @ cell + ((ix a1 + iy a2) + iz a3)`` — the very expression the reference's replication kernel evaluates
(src/repeat_cell.cpp:41-59), so knife-edge cutoffs see identical coordinates.  Miller-index orientations and ordered
multi-species structures are outside the hot path."""
import numpy as np

from . import kernels
from .box import Box
from .frame import Frame
from .parallel import get_num_threads

# fractional coordinates of the basis atoms
_FCC_SITES = [[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]]
_BASIS = {
    "sc": [[0.0, 0.0, 0.0]],
    "bcc": [[0.0, 0.0, 0.0], [0.5, 0.5, 0.5]],
    "fcc": _FCC_SITES,
    # fcc + the same lattice shifted by a quarter of the body diagonal (written out: the order of the sites is part of the output)
    "diamond": _FCC_SITES + [[0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]],
}
_HEX_BASIS = {"hcp": [[0.0, 0.0, 0.0], [1.0 / 3.0, 2.0 / 3.0, 0.5]], "graphene": [[0.0, 0.0, 0.0], [1.0 / 3.0, 2.0 / 3.0, 0.0]]}


def unit_cell(structure, a, c=None):
    """(3x3 cell with the vectors as rows, fractional basis) of a structure at lattice constant a (and c for the hexagonal ones)"""
    kind = structure.lower()
    if kind in _BASIS:
        return a * np.eye(3), np.array(_BASIS[kind], dtype=np.float64)
    if kind in _HEX_BASIS:
        if c is None:
            if kind == "graphene":
                raise ValueError("graphene needs c (the spacing of the periodic images along z)")
            c = a * float(np.sqrt(8 / 3))  # ideal close packing
        hexagonal = np.array([[a, 0.0, 0.0], [-0.5 * a, 0.5 * np.sqrt(3.0) * a, 0.0], [0.0, 0.0, c]])
        return hexagonal, np.array(_HEX_BASIS[kind])
    raise ValueError(f"Unrecognized structure '{structure}'. Supported here: {sorted(_BASIS) + ['hcp', 'graphene']}")


def _supercell(cell, nx, ny, nz):
    return cell * np.array([[nx], [ny], [nz]])


def lattice_positions(structure, a, nx=1, ny=1, nz=1, c=None):
    """(positions (N, 3), box (3, 3)) of an nx x ny x nz supercell, evaluated with numpy — for test inputs and checks; build_crystal itself
    always runs the replication kernel"""
    cell, basis = unit_cell(structure, a, c)
    sites = basis @ cell
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    shift = (ix[..., None] * cell[0] + iy[..., None] * cell[1]) + iz[..., None] * cell[2]
    pos = (shift[:, :, :, None, :] + sites[None, None, None, :, :]).reshape(-1, 3)
    # (sites + shift) in the kernel: addition commutes exactly, the grouping of the shift is what matters
    return np.ascontiguousarray(pos), _supercell(cell, nx, ny, nz)


def build_crystal(name, structure, a, miller1=None, miller2=None, miller3=None, nx=1, ny=1, nz=1, c=None):
    """``System`` holding an nx x ny x nz supercell of one element, the reference's positional order
    (src/mdapy/build_lattice.py:657-668).  The generator of this package's benchmarks and tests: standard orientation only
    (``miller1..3`` None or the unit vectors) — re-oriented cells, multi-species structures and the HEA / dislocation builders
    of the reference are outside the hot path (out of scope)."""
    from .system import System

    for k, (m, unit) in enumerate(zip((miller1, miller2, miller3), ((1, 0, 0), (0, 1, 0), (0, 0, 1)))):
        if m is not None and tuple(int(v) for v in m) != unit:
            raise ValueError(f"mdapy_amd.build_crystal builds the standard orientation only (miller{k + 1}={m!r})")

    if not isinstance(name, str):
        raise TypeError("only single-element crystals are supported here; pass one element symbol")
    cell, basis = unit_cell(structure, a, c)
    sites = np.ascontiguousarray(basis @ cell)
    flat = np.zeros(len(sites) * nx * ny * nz * 3, dtype=np.float64)
    kernels.repeat_cell.repeat_cell(flat, cell, sites, nx, ny, nz, get_num_threads())  # HIP kernel; raises without a GPU
    pos = flat.reshape((-1, 3))
    frame = Frame({"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2], "element": np.full(len(pos), name, dtype=object)})
    return System(data=frame, box=Box(_supercell(cell, nx, ny, nz)))
