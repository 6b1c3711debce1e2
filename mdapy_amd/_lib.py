"""ctypes binding of libmdapy_amd.so (the C ABI declared in include/mdapy_amd.h).

The reference binds its C++ with nanobind (CMakeLists.txt:71-100); nanobind is
not available in this image, so the same entry points are reached through
ctypes.  There is NO CPU fallback: if the shared library is missing, or no HIP
device is visible when a kernel is requested, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmdapy_amd.so")

HOST, DEVICE = 0, 1
ERR_BOX, ERR_HIP, ERR_ARG, ERR_NOMEM = -1, -2, -3, -4

_lib = None

i64, dbl, cint, vp = C.c_int64, C.c_double, C.c_int, C.c_void_p
# mdh_alloc_rows_fn (include/mdapy_amd.h): int (*)(void *user, int64_t N, int64_t M, int **verlet, double **dist)
ALLOC_ROWS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_double)))

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/mdapy_amd.h
_SIGNATURES = {
    "mdh_last_error": [],
    "mdh_version": [],
    "mdh_device_count": [],
    "mdh_set_device": [cint],
    "mdh_warm": [],
    "mdh_min_max_i32": [vp, i64, vp, cint, vp],
    "mdh_dense_codes_i32": [vp, i64, cint, cint, vp, vp, cint, vp],
    "mdh_release_workspace": [],
    "mdh_workspace_bytes": [],
    "mdh_prof_enable": [cint],
    "mdh_prof_reset": [],
    "mdh_prof_report": [vp, cint],
    "mdh_debug_set_neighbor_variant": [cint],
    "mdh_debug_set_indirect": [cint],
    "mdh_debug_neighbor_plan": [vp],
    "mdh_debug_set_fcna_variant": [cint],
    "mdh_debug_track_counters": [cint],
    "mdh_debug_counters": [vp],
    "mdh_debug_set_rdf_variant": [cint],
    "mdh_debug_set_knn_variant": [cint],
    "mdh_debug_set_sq_variant": [cint],
    "mdh_debug_set_entropy_variant": [cint],
    "mdh_debug_set_ptm_order_cap": [cint],
    "mdh_debug_image_thresholds": [dbl, vp],
    "mdh_parse_table": [vp, i64, cint, i64, cint, vp, vp, vp, i64, vp, cint, vp],
    "mdh_debug_text_pow5": [cint, vp],
    "mdh_debug_parse_double": [C.c_char_p, i64, vp],
    "mdh_build_neighbor": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, vp, i64, cint, cint, vp],
    "mdh_slab_halo_select": [vp, vp, vp, i64, vp, vp, dbl, dbl, vp, vp, vp, vp, vp, vp, i64, cint, vp],
    "mdh_hint_cell_window": [cint, dbl, dbl],
    "mdh_hint_centre_window": [cint, dbl, dbl],
    "mdh_cell_window_check": [vp],
    "mdh_slab_halo_messages": [vp, vp, vp, i64, vp, vp, dbl, dbl, vp, vp, cint, vp, vp, i64, vp],
    "mdh_slab_append_ghosts": [vp, vp, i64, i64, i64, vp, cint, vp, i64, vp],
    "mdh_order_statistic": [vp, vp, vp, i64, vp, vp, vp, vp, cint, vp],
    "mdh_spatial_sort": [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, cint, vp],
    "mdh_permute": [vp, vp, i64, cint, cint, vp, cint, vp],
    "mdh_gather_positions": [vp, vp, vp, vp, i64, vp, vp, vp, cint, vp],
    "mdh_translate_rows": [vp, vp, vp, vp, i64, i64, vp, vp, vp, cint, vp],
    "mdh_slab_append_ghosts_static": [vp, vp, i64, vp, cint, vp, i64, vp],
    "mdh_slab_overflow_check": [],
    "mdh_build_neighbor_keyed": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, vp, i64, cint, vp, cint, vp],
    "mdh_build_neighbor_fcna": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, vp, i64, cint, vp, vp, cint, vp],
    "mdh_neighbor_count": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, cint, vp],
    "mdh_build_neighbor_exact": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, ALLOC_ROWS, vp, cint, vp],
    "mdh_build_neighbor_exact_keyed": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, ALLOC_ROWS, vp, vp, cint, vp],
    "mdh_build_neighbor_exact_fcna": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, vp, ALLOC_ROWS, vp, vp, vp, cint, vp],
    "mdh_sort_verlet_by_distance": [vp, vp, i64, i64, cint, cint, vp],
    "mdh_wrap_positions": [vp, vp, vp, i64, vp, vp, vp, cint, vp],
    "mdh_average_by_neighbor": [dbl, vp, vp, vp, i64, i64, vp, vp, cint, cint, vp],
    "mdh_fcna": [vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, dbl, cint, vp],
    "mdh_acna": [vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, cint, vp],
    "mdh_ids": [vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, cint, vp],
    "mdh_csp": [vp, vp, vp, i64, vp, vp, vp, vp, i64, cint, vp, cint, vp],
    "mdh_get_sq": [vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, cint, cint, cint, cint, cint, cint, cint,
                   dbl, cint, vp, vp, vp, cint, vp],
    "mdh_identify_solid_liquid": [cint, vp, vp, vp, vp, i64, i64, vp, vp, cint, cint, dbl, cint, vp, vp, cint, cint,
                                  dbl, cint, vp],
    "mdh_rdf": [vp, vp, vp, vp, i64, i64, vp, cint, dbl, cint, cint, vp],
    "mdh_rdf_single_species": [vp, vp, vp, i64, i64, vp, dbl, cint, cint, vp],
    "mdh_rdf_streaming": [vp, vp, vp, vp, i64, vp, vp, vp, vp, cint, dbl, cint, cint, vp],
    "mdh_wcp": [vp, vp, vp, i64, i64, cint, vp, cint, vp],
    "mdh_wcp_counts": [vp, vp, vp, vp, i64, i64, cint, vp, cint, vp],
    "mdh_knn": [vp, vp, vp, i64, vp, vp, vp, cint, vp, vp, cint, vp],
    "mdh_knn_keyed": [vp, vp, vp, i64, vp, vp, vp, cint, vp, vp, vp, cint, vp],
    "mdh_knn_keyed_rows": [vp, vp, vp, i64, vp, vp, vp, cint, vp, vp, vp, vp, vp, cint, vp, cint, vp],
    "mdh_knn_rows_width": [cint],
    "mdh_repeat_cell": [vp, vp, vp, i64, cint, cint, cint, cint, vp],
    "mdh_ptm": [C.c_char_p, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, dbl, vp, cint, vp, cint, cint, vp],
    "mdh_ptm_flags": [C.c_char_p],
    "mdh_aja": [vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, cint, vp],
    "mdh_cnp": [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, dbl, cint, vp],
    "mdh_structure_entropy": [dbl, dbl, cint, dbl, vp, vp, i64, i64, vp, cint, vp],
    "mdh_atomic_temperature": [vp, vp, i64, i64, vp, vp, vp, vp, vp, dbl, cint, vp],
    "mdh_cluster": [vp, vp, vp, i64, i64, dbl, cint, vp, vp, cint, vp],
    "mdh_identify_sftb_fcc": [vp, i64, vp, vp, vp, i64, vp, cint, cint, vp],
    "mdh_filter_overlap_atom": [vp, vp, vp, i64, vp, vp, vp, dbl, vp, cint, vp],
    "mdh_voronoi_volume_number_radius": [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, cint, vp],
    "mdh_voronoi_neighbor_count": [vp, vp, vp, i64, vp, vp, vp, vp, vp, cint, vp],
    "mdh_voronoi_neighbor": [vp, vp, vp, i64, vp, vp, vp, dbl, dbl, vp, vp, vp, cint, cint, vp],
    "mdh_voronoi_neighbor_rows": [vp, vp, vp, i64, vp, vp, vp, dbl, dbl, vp, vp, vp, cint, vp, vp, cint, vp],
    "mdh_filter_overlap_atom_with_grain": [vp, vp, vp, vp, vp, i64, vp, vp, vp, dbl, dbl, dbl, vp, cint, vp],
    "mdh_transform_and_filter": [vp, vp, vp, i64, vp, vp, vp, vp, cint, vp, vp, cint, vp],
    "mdh_voronoi_cell_info": [vp, vp, vp, i64, vp, vp, vp, cint, cint, vp, vp, vp, vp, vp, vp, vp, cint, vp],
    "mdh_voronoi_row_distance": [vp, i64, cint, vp, vp, vp, vp, vp, vp, vp, cint, vp],
    "mdh_sfc_direct": [vp, vp, vp, i64, vp, vp, cint, dbl, dbl, vp, vp, vp, i64, C.c_uint, cint, vp],
    "mdh_sfc_direct_partial": [vp, vp, vp, vp, cint, i64, vp, vp, cint, dbl, dbl, cint, vp],
    "mdh_filter_by_type": [vp, vp, vp, vp, i64, i64, vp, vp, vp, cint, cint, vp],
}
_RESTYPES = {"mdh_last_error": C.c_char_p, "mdh_workspace_bytes": C.c_int64}

EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mdapy_amd/csrc`).  mdapy_amd has no CPU fallback."
            )
        # torch ships its own libamdhip64.so.7; a process must not end up with two HIP runtimes (the second
        # one to initialise reports "No HIP GPUs are available").  Importing torch first makes the dynamic
        # linker resolve our library's libamdhip64.so.7 to the copy torch already loaded.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is only plumbing (device memory, streams); the C ABI also works without it
            pass
        L = C.CDLL(LIB_PATH)
        for name, argt in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = _RESTYPES.get(name, C.c_int)
        # MDAPY_HIP_DEVICE ("Config / flags"): the device the library's calls of this process go to, chosen
        # once at load; mdh_set_device() switches it later (bench.py: one process per GPU, LOCAL_RANK)
        want = os.environ.get("MDAPY_HIP_DEVICE", "").strip()
        if want:
            try:
                index = int(want)
            except ValueError:
                raise ValueError(f"MDAPY_HIP_DEVICE must be a device index, got {want!r}") from None
            count = int(L.mdh_device_count())
            if count > 0:  # (a CPU-only box: nothing to select; compute calls raise on their own)
                if not 0 <= index < count:
                    raise ValueError(f"MDAPY_HIP_DEVICE={index}: this process sees {count} HIP device(s)")
                if L.mdh_set_device(index) != 0:
                    raise RuntimeError(L.mdh_last_error().decode("utf-8", "replace"))
        # The code objects of all kernel families, loaded now (mdh_warm): the set-up cost of a process belongs to its first
        # touch of the library — System(...) — not to its first build_neighbor / cal_* (tools/cold_path.py).  Not under a
        # multi-process launcher (LOCAL_RANK set, no MDAPY_HIP_DEVICE): the process has not selected its GPU yet, and the
        # library warms the right one by itself at its first call there (runtime.hip Scope) or in mdh_set_device.
        # MDAPY_HIP_WARM=0 switches all of it off.
        if os.environ.get("MDAPY_HIP_WARM", "1") != "0" and (want or "LOCAL_RANK" not in os.environ) and int(L.mdh_device_count()) > 0:
            if L.mdh_warm() != 0:
                raise RuntimeError(L.mdh_last_error().decode("utf-8", "replace"))
        _lib = L
    return _lib


def device_count() -> int:
    return int(lib().mdh_device_count())


def check(rc: int) -> None:
    """Translate a C-ABI return code into the exception the reference would raise."""
    if rc == 0:
        return
    msg = lib().mdh_last_error().decode("utf-8", "replace")
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    # ERR_BOX mirrors the C++ throw in src/box.h:185-186, which nanobind turns into RuntimeError
    raise RuntimeError(msg)


def same_rows(what: str, n_atoms: int, **arrays) -> None:
    """Every per-atom argument of one call must have one row per atom.  The reference indexes them with the length of
    ``x`` without a check (e.g. src/centro_symmetry_parameter.cpp:24: out-of-bounds reads when a stale list is passed);
    the drop-in refuses instead."""
    for name, a in arrays.items():
        if a is None:
            continue
        rows = int(a.shape[0]) if hasattr(a, "shape") else len(a)
        if rows != int(n_atoms):
            raise ValueError(f"{what}: {name} has {rows} rows for {int(n_atoms)} atoms")


_box_memo = {}  # bytes of (box, origin, boundary) -> the converted copies and their pointers (a trajectory asks for the same box again and again)


def host_box(box, origin, boundary):
    """(box9, origin3, boundary3) as C-contiguous host arrays + their pointers (kept alive by the caller)."""
    if type(box) is np.ndarray and type(origin) is np.ndarray and type(boundary) is np.ndarray and box.dtype == np.float64 and origin.dtype == np.float64:
        key = (box.tobytes(), origin.tobytes(), boundary.tobytes(), boundary.dtype.str)
        hit = _box_memo.get(key)
        if hit is not None:
            return hit
    else:
        key = None
    b = np.array(np.asarray(box, dtype=np.float64).reshape(3, 3), order="C", copy=True)
    o = np.array(np.asarray(origin, dtype=np.float64).reshape(3), order="C", copy=True)
    # the reference accepts int64 boundary through nanobind's implicit conversion
    p = np.ascontiguousarray(np.asarray(boundary).astype(np.int32).reshape(3))
    out = ((b, o, p), (b.ctypes.data, o.ctypes.data, p.ctypes.data))
    if key is not None and len(key[0]) == 72 and len(key[1]) == 24:
        if len(_box_memo) > 64:
            _box_memo.clear()
        _box_memo[key] = out  # (private copies: the caller's arrays may change under us, the memo is keyed by content)
    return out
