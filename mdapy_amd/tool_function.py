"""Host-side helpers shared by the analysis classes; mirrors the part of
src/mdapy/tool_function.py that the hot path uses (:75-192)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _neighbor, _repeat_cell
from .box import Box
from .devarray import as_numpy, zeros
from .frame import Frame, concat
from .parallel import get_num_threads


def xyz(data: Frame):
    """the three position columns (frame columns carry their HBM mirror)"""
    return data["x"], data["y"], data["z"]


def sort_neighbor(verlet_list, distance_list, neighbor_number, k: int) -> None:
    """Sort the first ``k`` neighbours of every atom by distance, in place (tool_function.py:75-119)."""
    minNumber = neighbor_number.min()
    assert minNumber >= k, f"The min neighbor number {minNumber} is lower than k {k}."
    _neighbor.sort_verlet_by_distance(verlet_list, distance_list, k, get_num_threads())


def wrap_pos(data: Frame, box: Box) -> Frame:
    """tool_function.py:122-138"""
    x, y, z = (data[c].to_numpy(writable=True) for c in "xyz")
    _neighbor.wrap_positions(x, y, z, box.box, box.origin, box.boundary, get_num_threads())
    return data.with_columns(x=x, y=y, z=z)


def _tile_positions(data: Frame, box: Box, nx: int, ny: int, nz: int) -> Tuple[np.ndarray, np.ndarray]:
    old_pos = np.ascontiguousarray(data.select("x", "y", "z").to_numpy(), dtype=np.float64)
    total = old_pos.shape[0] * nx * ny * nz * 3
    new_pos = np.zeros(total, dtype=np.float64)
    _repeat_cell.repeat_cell(new_pos, box.box, old_pos, nx, ny, nz, get_num_threads())
    return new_pos.reshape((-1, 3)), box.box * np.array([nx, ny, nz]).reshape((3, 1))


def replicate(data: Frame, box: Box, nx: int, ny: int, nz: int) -> Tuple[Frame, Box]:
    """Replicate all columns nx*ny*nz times, cell-major with the original atoms first
    (tool_function.py:141-176); an ``id`` column is renumbered from 1."""
    nx, ny, nz = int(nx), int(ny), int(nz)
    new_pos, new_box = _tile_positions(data, box, nx, ny, nz)
    new = concat([data] * (nx * ny * nz)).with_columns(x=new_pos[:, 0], y=new_pos[:, 1], z=new_pos[:, 2])
    if "id" in new.columns:
        new = new.with_columns(id=np.arange(1, new.shape[0] + 1, dtype=np.asarray(data["id"]).dtype))
    return new, Box(new_box, box.boundary, box.origin)


def dense_labels(labels) -> Tuple[list, np.ndarray]:
    """(sorted unique labels as Python objects, index of every entry in that list as int32) — what the reference builds with
    ``sorted(set(x.tolist()))`` and a per-atom dictionary lookup (e.g. radial_distribution_function.py:136-142), without the
    per-atom Python loop: one comparison pass when all atoms carry the same label, ``np.unique`` otherwise."""
    raw = np.asarray(labels)
    if raw.size == 0:
        return [], np.zeros(0, np.int32)
    first = raw.flat[0]
    if bool((raw == first).all()):
        return [first.item() if hasattr(first, "item") else first], np.zeros(raw.shape[0], np.int32)
    uniq, inv = np.unique(raw, return_inverse=True)
    return uniq.tolist(), inv.reshape(-1).astype(np.int32)


def _replicate_pos(data: Frame, box: Box, nx: int, ny: int, nz: int) -> Tuple[Frame, Box]:
    """positions only (tool_function.py:179-192)"""
    nx, ny, nz = int(nx), int(ny), int(nz)
    new_pos, new_box = _tile_positions(data, box, nx, ny, nz)
    return Frame({"x": new_pos[:, 0], "y": new_pos[:, 1], "z": new_pos[:, 2]}), Box(new_box, box.boundary, box.origin)


def average_by_neighbor(average_rc: float, data: Frame, verlet_list, distance_list, neighbor_number,
                        property_name: str, include_self: bool = True, output_name: Optional[str] = None) -> Frame:
    """tool_function.py:19-72"""
    assert property_name in data.columns
    N = data.shape[0]
    out = zeros(N, np.float64)
    _neighbor.average_by_neighbor(average_rc, verlet_list, distance_list, neighbor_number, data[property_name], out,
                                  include_self, get_num_threads())
    if output_name is None:
        output_name = f"{property_name}_ave"
    return data.with_columns(**{output_name: as_numpy(out)})
