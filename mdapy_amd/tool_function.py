"""Helpers with the names and signatures of src/mdapy/tool_function.py:19-192 — the part of that module the hot path
and its callers use.  The work itself is in :mod:`mdapy_amd.policy` and the kernels."""
import numpy as np

from . import kernels, policy
from .devarray import as_numpy, zeros
from .parallel import get_num_threads

dense_labels = policy.label_codes
xyz = policy.positions


def sort_neighbor(verlet_list, distance_list, neighbor_number, k):
    """the k nearest of every row to the front, ordered by distance (in place); every atom needs >= k neighbours"""
    fewest = neighbor_number.min()
    if fewest < k:
        raise AssertionError(f"The min neighbor number {fewest} is lower than k {k}.")
    kernels.neighbor.sort_verlet_by_distance(verlet_list, distance_list, k, get_num_threads())


def wrap_pos(data, box):
    """frame with all positions folded back into the box along its periodic directions"""
    coords = {c: data[c].to_numpy(writable=True) for c in ("x", "y", "z")}
    kernels.neighbor.wrap_positions(coords["x"], coords["y"], coords["z"], *policy.box_args(box), get_num_threads())
    return data.with_columns(**coords)


def replicate(data, box, nx, ny, nz):
    """(frame, box) of nx * ny * nz copies, every column carried along, originals first"""
    return policy.replica(data, box, (nx, ny, nz), all_columns=True)


def _replicate_pos(data, box, nx, ny, nz):
    """the same with positions only"""
    return policy.replica(data, box, (nx, ny, nz), all_columns=False)


def average_by_neighbor(average_rc, data, property_name, verlet_list, distance_list, neighbor_number, include_self=True,
                        output_name=None):  # (src/mdapy/tool_function.py:14-23: this positional order)
    """frame with ``<property>_ave`` (or ``output_name``): mean of a column over the neighbours within ``average_rc``"""
    assert property_name in data.columns
    mean = zeros(data.shape[0], np.float64)
    kernels.neighbor.average_by_neighbor(average_rc, verlet_list, distance_list, neighbor_number, data[property_name], mean,
                                         include_self, get_num_threads())
    name = f"{property_name}_ave" if output_name is None else output_name
    return data.with_columns(**{name: as_numpy(mean)})
