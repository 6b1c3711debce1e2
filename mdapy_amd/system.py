"""``System`` — the user-facing object of the hot path, with the surface of ``mdapy.System`` (src/mdapy/system.py) above
the neighbor list and the per-atom structural analyses.

A system is a frame of per-atom columns plus a box.  It remembers ONE neighbor list at a time — ``verlet_list``,
``distance_list``, ``neighbor_number`` and, for a cutoff list, its ``rc`` (system.py:1155-1166) — and, when that list
was built on a replica of a thin periodic box, the replica (``_enlarge_data`` / ``_enlarge_box``) its indices refer to.
Assigning a box or replacing the data with ``reset_neighbor=True`` forgets the list (system.py:232-245, 748-763).

Every ``cal_*`` method first makes sure a suitable list exists, following the reference's reuse rules (SURVEY.md 3,
Appendix C), then runs its analysis class and stores the first N rows of the result as columns of ``data`` or returns
the result object.  The reuse rules are three helpers here — ``_require_cutoff_list``, ``_nearest_prefix`` and
``_borrow_nearest`` — instead of being restated in every method.

The lists stay in HBM between calls (:class:`mdapy_amd.devarray.HArray`) and turn into numpy on first host access.
"""
import numpy as np

from . import policy
from . import tool_function as tool
from .ackland_jones_analysis import AcklandJonesAnalysis
from .atomic_temperature import AtomicTemperature
from .box import Box
from .centro_symmetry_parameter import CentroSymmetryParameter
from .cluster_analysis import ClusterAnalysis
from .common_neighbor_analysis import CommonNeighborAnalysis
from .common_neighbor_parameter import CommonNeighborParameter
from .devarray import HArray, as_numpy, full
from .frame import Frame
from .identify_diamond_structure import IdentifyDiamondStructure
from .identify_fcc_planar_faults import IdentifyFccPlanarFaults
from .knn import NearestNeighbor
from .neighbor import Neighbor
from .polyhedral_template_matching import PolyhedralTemplateMatching
from .radial_distribution_function import RadialDistributionFunction
from .steinhardt_bond_orientation import SteinhardtBondOrientation
from .structure_entropy import StructureEntropy
from .structure_factor import StructureFactor
from .voronoi import Voronoi
from .warren_cowley_parameter import WarrenCowleyParameter

_REPLICA = ("_enlarge_box", "_enlarge_data")
_LIST = ("verlet_list", "neighbor_number", "distance_list", "rc", "_sorted_columns", "_list_cutoff") + _REPLICA


# ----------------------------------------------------------------------------------------------------------------------
# The cell-sorted twin.  The reference's kernels do not care in which order atoms arrive (a linked list per cell,
# src/neighbor.cpp:64-100); a GPU's gathers do: on an id-sorted dump of a diffused system, or a shuffled one, a neighbour's
# position is an HBM access instead of an L2 hit, and the fixed-cutoff CNA of 10 M such atoms took 6.7 ms instead of 0.54
# (profiles/r05_order_sweep.txt).  A large system that was handed in in no spatial order therefore gets a TWIN: the same atoms
# in cell order (csrc/order.hip), with every other column read through the permutation.  List builds and the analyses that do
# not depend on atom numbering run on the twin — lists keyed by the original index, so rows come out in the reference's order
# and every sum runs over the same numbers in the same order — and what the user reads is translated back: per-atom columns
# by one scatter, the rows of a list only if somebody asks for them (devarray.LazyHArray).  MDAPY_SPATIAL_SORT = 0 (never),
# 1 (always, whatever the order looks like; any size — tests), unset (systems of MDAPY_SORT_MIN_ATOMS atoms and more whose
# order statistic says so).
# ----------------------------------------------------------------------------------------------------------------------
import functools
import os

SORT_MIN_ATOMS = int(os.environ.get("MDAPY_SORT_MIN_ATOMS", "200000"))
SORT_FAR_FRACTION = 0.25  # of consecutive atoms in bins that do not touch (mdh_order_statistic): a lattice builder's order has < 0.01


def _on_twin(method):
    """run a System method on the cell-sorted twin when there is one (and the call's arguments allow it); copy back what it left"""
    name = method.__name__

    @functools.wraps(method)
    def call(self, *args, **kwargs):
        twin = self._twin_for(name, args, kwargs)
        if twin is None:
            return method(self, *args, **kwargs)
        return self._run_on_twin(twin, name, args, kwargs)

    return call


def _dev_of(harray):
    return harray.dev()


# positional parameter names of the methods that may run on the twin (what _twin_for looks at)
_ARGS = {
    "build_neighbor": ("rc", "max_neigh"),
    "build_nearest_neighbor": ("k",),
    "cal_common_neighbor_analysis": ("rc", "max_neigh"),
    "cal_common_neighbor_parameter": ("rc", "max_neigh"),
    "cal_structure_entropy": ("rc", "sigma", "use_local_density", "average_rc", "max_neigh"),
    "cal_atomic_temperature": ("rc", "factor", "max_neigh"),
    "cal_steinhardt_bond_orientation": ("llist", "use_voronoi", "nnn", "rc", "average", "use_weight", "weight", "wl", "wlhat",
                                        "a_face_area_threshold", "r_face_area_threshold", "identify_liquid"),
    "cal_radial_distribution_function": ("rc", "nbin", "max_neigh", "streaming"),
    "cal_warren_cowley_parameter": ("rc", "max_neigh"),
    "average_by_neighbor": ("average_rc", "property_name"),
}


def _position_columns(xyz):
    """x, y, z columns of an (N, 3) array.  With a GPU the array crosses PCIe once, as it is, and is split into columns in HBM
    (the columns' host copies are made if somebody asks for them): the three strided host copies of the reference's
    pos[:, k] pattern were 70 of the 90 ms a 10 M-atom numpy-in / labels-out call took."""
    from .devarray import have_gpu

    if have_gpu() and xyz.shape[0] >= (1 << 16):
        from .devarray import HArray, torch

        t = torch()
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (a read-only source array: the tensor is only read)
            dev = t.from_numpy(np.ascontiguousarray(xyz)).to("cuda")
        cols = dev.t().contiguous()  # one transposing pass on the device
        return {name: HArray(cols[k]) for k, name in enumerate("xyz")}
    return dict(zip("xyz", xyz.T))


def _from_ase(atoms):
    """frame and box of an ASE ``Atoms`` (load_save.py:508-545): cell rows, ``pbc``, positions, chemical symbols.  Duck-typed:
    ase itself is not needed, only the four getters the reference calls."""
    for needed in ("get_cell", "get_pbc", "get_positions", "get_chemical_symbols"):
        if not hasattr(atoms, needed):
            raise TypeError("Only accept an ASE Atoms object")
    cell = Box(np.array(atoms.get_cell(), dtype=np.float64), [1 if p else 0 for p in atoms.get_pbc()])
    xyz = np.asarray(atoms.get_positions(), dtype=np.float64)
    cols = _position_columns(xyz)
    cols["element"] = np.array(atoms.get_chemical_symbols())
    return Frame(cols), cell


def _from_ovito(collection):
    """frame, box and attributes of an OVITO ``DataCollection`` (load_save.py:413-505): the 3x4 cell transposed (vectors + origin),
    ``Position`` -> x y z, ``Particle Type`` -> type, ``Particle Identifier`` -> id, ``Velocity`` / ``Force`` -> three columns,
    ``Velocity Magnitude`` dropped, every other property under its name without blanks (vectors as name_0, name_1, ...);
    an ``element`` column when every particle type carries a name.  Duck-typed: ``.cell`` (array-like with ``.pbc``),
    ``.particles`` (mapping), ``.attributes`` (mapping)."""
    for needed in ("cell", "particles", "attributes"):
        if not hasattr(collection, needed):
            raise TypeError("Only accept an Ovito DataCollection object")
    # the whole 3x4 cell transposed: rows 0-2 the vectors, row 3 the origin (load_save.py:443-444)
    cell = Box(np.array(collection.cell[...], dtype=np.float64).T, [1 if p else 0 for p in collection.cell.pbc])
    info = {key: value for key, value in collection.attributes.items()}
    three = {"Position": ("x", "y", "z"), "Velocity": ("vx", "vy", "vz"), "Force": ("fx", "fy", "fz")}
    one = {"Particle Type": "type", "Particle Identifier": "id"}
    cols = {}
    for key in collection.particles.keys():
        values = np.array(collection.particles[key][...])
        if key in three:
            for k, name in enumerate(three[key]):
                cols[name] = np.ascontiguousarray(values[:, k])
        elif key in one:
            cols[one[key]] = values
        elif key == "Velocity Magnitude":
            continue
        else:
            name = "".join(key.split())
            if values.ndim == 1:
                cols[name] = values
            else:
                for k in range(values.shape[1]):
                    cols[f"{name}_{k}"] = np.ascontiguousarray(values[:, k])
    table = getattr(collection.particles, "particle_type", None)
    if table is not None and "type" in cols:
        names = {t.id: t.name for t in table.types}
        if names and all(isinstance(n, str) and len(n) > 0 for n in names.values()):
            cols["element"] = np.array([names[t] for t in cols["type"].tolist()])
    return Frame(cols), cell, info


# The frames of a trajectory share their atom numbering: when a System has just been sorted and the next one brings as many atoms in
# the same box shape, its positions are first read through the LAST permutation — atoms move a fraction of a cell between frames, the
# old cell order is still a spatial order — and the order statistic of the result decides whether that will do (three gathers and a
# sampling pass instead of a sort: the sort is as long as the whole neighbor + CNA step of an ordered frame).
_last_order = {}  # "perm": HArray / ndarray, "n": atoms, "pbc": tuple


def _remember_order(perm, n, where):
    _last_order.update(perm=perm, n=int(n), pbc=tuple(int(v) for v in np.asarray(where[5]).ravel()), host=isinstance(perm, np.ndarray))


def _sorted_as_last_time(cols, where, n):
    from . import kernels

    perm = _last_order.get("perm")
    if perm is None or _last_order.get("n") != int(n) or os.environ.get("MDAPY_REUSE_ORDER", "1") == "0":
        return None
    if _last_order.get("pbc") != tuple(int(v) for v in np.asarray(where[5]).ravel()):
        return None
    on_dev = any(c._host_arr is None or c._dev is not None for c in cols)
    if on_dev == bool(_last_order.get("host")):
        return None  # (the permutation lives in the other memory space)
    try:
        arrs = [c.device_array() if on_dev else c.to_numpy() for c in cols]
        if hasattr(kernels.order, "gather_positions"):
            moved = list(kernels.order.gather_positions(*arrs, perm))
        else:
            moved = [kernels.order.permute(a, perm) for a in arrs]
        if kernels.order.order_statistic(*moved, *where[3:]) > SORT_FAR_FRACTION:
            return None  # another numbering after all: sort
    except Exception:
        return None
    return moved[0], moved[1], moved[2], perm, int(n)


class System:
    def __init__(self, filename=None, data=None, pos=None, box=None, ase_atom=None, ovito_atom=None, format=None,
                 global_info=None):
        self._info = {}
        self._sort_mode = os.environ.get("MDAPY_SPATIAL_SORT", "")  # (fixed when the system is made: see _spatial)
        if isinstance(filename, str):
            from .load_save import read_file

            self._frame, self.box, self._info = read_file(filename, format)
        elif box is not None and data is not None:
            self._frame = Frame.from_any(data)
            for name in ("x", "y", "z"):
                if name not in self._frame.columns:
                    raise AssertionError(f"data must contain column {name!r}.")
            self.box = box
        elif box is not None and pos is not None:
            xyz = np.asarray(pos, dtype=np.float64)
            if xyz.ndim != 2 or xyz.shape[1] != 3:
                raise AssertionError("pos must have shape (N, 3).")
            self._frame = Frame(_position_columns(xyz))
            self.box = box
        elif ase_atom is not None:
            self._frame, self.box = _from_ase(ase_atom)
        elif ovito_atom is not None:
            self._frame, self.box, self._info = _from_ovito(ovito_atom)
        else:
            raise RuntimeError("One must at least provide filename or [data, box] or [pos, box] or ase_atom or ovito_atom.")
        if global_info is not None and not self._info:
            self._info = dict(global_info)

    # ---------------------------------------------------------------- state
    def _forget(self, names):
        for name in names:
            self.__dict__.pop(name, None)
        if "verlet_list" in names:  # the list is gone: the twin's too, and what mirrored it
            state = self.__dict__.get("_twin_state")
            if state is not None and state[1] is not None:
                state[1]._forget(_LIST)
            self.__dict__.pop("_mirror", None)

    # ------------------------------------------------------- the cell-sorted twin (see the top of the module)
    def _spatial(self):
        """the twin, made on first use; None: this system is analysed in the order it has"""
        if self.__dict__.get("_is_twin"):
            return None
        mode = self.__dict__.get("_sort_mode", "")
        if mode == "0":
            return None
        cols = tuple(self._frame[c] for c in ("x", "y", "z"))
        # (the twin belongs to THESE column objects and THIS box: the state keeps them alive, so that "the same object" cannot be a
        # new one at a recycled address)
        key = (*cols, self._cell)
        state = self.__dict__.get("_twin_state")
        if state is not None and len(state[0]) == len(key) and all(a is b for a, b in zip(state[0], key)):
            return state[1]
        twin = None
        from .devarray import have_gpu
        from . import kernels

        big = self.N >= SORT_MIN_ATOMS or (mode == "1" and self.N >= 2)
        if big and hasattr(kernels, "order") and (have_gpu() or mode == "1") and all(np.dtype(c.dtype) == np.float64 for c in cols) \
                and policy.is_single(self._safe_repeat()):
            where = (*cols, *policy.box_args(self.box))
            if mode == "1" or kernels.order.order_statistic(*where) > SORT_FAR_FRACTION:
                xs, ys, zs, perm, n = _sorted_as_last_time(cols, where, self.N) or kernels.order.spatial_sort(*where)
                if n == self.N:
                    _remember_order(perm, self.N, where)
                    twin = System(data=Frame({"x": xs, "y": ys, "z": zs}), box=self.box)
                    twin._is_twin = True
                    twin._perm = perm
                    # the in-cell ordering key of the twin's list builds: the ORIGINAL index, so that a row lists its atoms in the
                    # order the reference would (descending index inside a cell, neighbor.cpp:97-98)
                    twin._order_key = (HArray(perm.dev().long()) if isinstance(perm, HArray) else np.asarray(perm, np.int64))
        self._twin_state = (key, twin)
        self.__dict__.pop("_mirror", None)
        self.__dict__.pop("_twin_cols", None)
        return twin

    def _twin_for(self, name, args, kwargs):
        """the twin if method ``name`` may run on it with these arguments"""
        twin = self._spatial()
        if twin is None:
            return None
        # the list this system remembers must be the twin's (translated), or neither has one
        mirror = self.__dict__.get("_mirror")
        mine = self.__dict__.get("verlet_list")
        if mine is not None and (mirror is None or mirror["rows"] is not mine):
            return None
        if mine is None and "verlet_list" in twin.__dict__:
            twin._forget(_LIST)
        bound = dict(zip(_ARGS.get(name, ()), args), **kwargs)
        reach = bound.get("rc", bound.get("average_rc"))
        if isinstance(reach, (int, float, np.integer, np.floating)) and reach > 0 and \
                not policy.is_single(policy.axis_copies(self.box, 2.0 * float(reach))):
            return None  # the build would search a replica: the ordering key cannot follow
        if name == "cal_steinhardt_bond_orientation" and (bound.get("use_voronoi") or bound.get("identify_liquid")
                                                         or bound.get("weight") is not None):
            return None  # (a caller's weight array lines up with THIS system's rows, not with the twin's permuted ones)
        return twin

    def _run_on_twin(self, twin, name, args, kwargs):
        from . import kernels
        from .devarray import LazyHArray
        from .frame import PermutedColumn

        perm = twin._perm
        # every column the twin does not have yet, read through the permutation (nothing moves before a kernel asks)
        cache = self.__dict__.setdefault("_twin_cols", {})
        cols = {}
        for cname in twin._frame.columns[:3]:
            cols[cname] = twin._frame[cname]
        for cname in self._frame.columns:
            if cname in ("x", "y", "z"):
                continue
            src = self._frame[cname]
            hit = cache.get(cname)
            if hit is None or hit[0] is not src:
                hit = cache[cname] = (src, PermutedColumn(src, perm))
            cols[cname] = hit[1]
        twin._frame = Frame(cols)
        twin._frame.order_key = twin._order_key  # (what a k-nearest search on this frame breaks exact ties by: knn.py)
        before = {cname: twin._frame[cname] for cname in twin._frame.columns}
        result = getattr(twin, name)(*args, **kwargs)
        # per-atom results: columns the call added or replaced, back in this system's order
        restored = {}
        for cname in twin._frame.columns:
            col = twin._frame[cname]
            if before.get(cname) is col:
                continue
            kind = np.dtype(col.dtype)
            if col._host_arr is None and kind.kind in "iuf" and kind.itemsize in (4, 8):
                restored[cname] = kernels.order.permute(col.device_array(), perm, scatter=True)
            else:
                out = np.empty_like(col.to_numpy())
                out[np.asarray(perm)] = col.to_numpy()
                restored[cname] = out
        if restored:
            self.update_data(self._frame.with_columns(**restored))
        for attr in ("cluster_number",):
            if attr in twin.__dict__:
                setattr(self, attr, twin.__dict__[attr])
        # the list the twin remembers now, as this system's: translated when somebody reads it
        self._mirror_lists(twin)
        if "ptm_indices" in twin.__dict__ and name == "cal_polyhedral_template_matching":
            src = twin.ptm_indices
            self.ptm_indices = LazyHArray(lambda: _dev_of(kernels.order.translate_rows(src, None, None, perm)[0]), src.shape, np.int32) \
                if isinstance(src, HArray) else kernels.order.translate_rows(np.asarray(src), None, None, np.asarray(perm))[0]
        if name == "build_neighbor" and result is not None:  # the labels of build_neighbor(..., _label=True): per atom, the twin's order
            result = kernels.order.permute(result, perm, scatter=True)
        return result

    def _mirror_lists(self, twin):
        from . import kernels
        from .devarray import LazyHArray

        if "verlet_list" not in twin.__dict__:
            for attr in _LIST:
                self.__dict__.pop(attr, None)
            self.__dict__.pop("_mirror", None)
            return
        rows, dist, counts = twin.verlet_list, twin.distance_list, twin.neighbor_number
        state = (rows, dist, counts, twin.__dict__.get("_sorted_columns", (None, 0))[1])  # (the objects themselves: see _spatial)
        mirror = self.__dict__.get("_mirror")
        same = mirror is not None and all(a is b for a, b in zip(mirror["state"][:3], state[:3])) and mirror["state"][3] == state[3]
        if not same or self.__dict__.get("verlet_list") is not mirror["rows"]:
            perm = twin._perm
            done = {}

            def translated(k):
                if not done:
                    done["v"], done["d"], done["n"] = kernels.order.translate_rows(rows, dist, counts, perm)
                return done[k]

            if isinstance(rows, HArray):
                out = (LazyHArray(lambda: _dev_of(translated("v")), rows.shape, np.int32),
                       LazyHArray(lambda: _dev_of(translated("d")), dist.shape, np.float64),
                       # (the counts alone: `_deep_enough` and the overflow check read them; the N x M rows need not move for that)
                       LazyHArray(lambda: _dev_of(done["n"] if done else kernels.order.permute(counts, perm, scatter=True)), counts.shape, np.int32))
            else:
                out = (translated("v"), translated("d"), translated("n"))
            mirror = self._mirror = {"state": state, "rows": out[0]}
            self.verlet_list, self.distance_list, self.neighbor_number = out
        for attr in ("rc", "_list_cutoff"):
            if attr in twin.__dict__:
                setattr(self, attr, twin.__dict__[attr])
            else:
                self.__dict__.pop(attr, None)
        self._sorted_columns = (id(self.verlet_list), twin.__dict__.get("_sorted_columns", (None, 0))[1])
        for attr in _REPLICA:
            self.__dict__.pop(attr, None)

    @property
    def box(self):
        return self._cell

    @box.setter
    def box(self, value):
        # Cartesian coordinates stay as they are; everything computed with the old box is gone
        self._cell = value if isinstance(value, Box) else Box(value)
        self._forget(_LIST)

    data = property(lambda self: self._frame)
    global_info = property(lambda self: self._info)
    N = property(lambda self: self._frame.shape[0])

    def __repr__(self):
        return f"Atom Number: {self.N}\n{self.box}\nParticle Information:\n{self.data}"

    def update_data(self, data, reset_calculator=False, reset_neighbor=False, reset_calcolator=None):
        """replace the per-atom frame; ``reset_neighbor`` also forgets the neighbor list.  ``reset_calcolator`` is the
        reference's deprecated misspelling of ``reset_calculator`` (system.py:686-744), accepted with the same warning; there
        is no calculator on this path, so neither flag has anything to clear."""
        if reset_calcolator is not None:
            import warnings

            warnings.warn("`reset_calcolator` is a misspelling and is deprecated; use `reset_calculator` instead.",
                          DeprecationWarning, stacklevel=2)
        self._frame = Frame.from_any(data)
        if reset_neighbor:
            self._forget(_LIST)

    def _get_compute_view(self):
        """(box, frame) the remembered list's indices refer to: the replica if there is one"""
        if "_enlarge_data" in self.__dict__:
            return self._enlarge_box, self._enlarge_data
        return self.box, self.data

    def _store(self, **columns):
        """results of an analysis over the compute view -> columns of the N real atoms"""
        n = self.N
        kept = {name: (values.head(n) if isinstance(values, HArray) and values.ndim == 1 else as_numpy(values)[:n])
                for name, values in columns.items()}  # results that are in HBM stay there until somebody reads them
        self.update_data(self._frame.with_columns(**kept))

    def wrap_pos(self):
        self.update_data(tool.wrap_pos(self._frame, self.box), reset_neighbor=True)

    # ---- small host helpers of the reference's System (src/mdapy/system.py:333-500, 786-850, 1414-1490): no kernel behind them
    def set_element(self, element):
        """element names: one string for every atom, or one name per atom (system.py:333-364)"""
        if isinstance(element, str):
            names = np.full(self.N, element, dtype=object)
        else:
            assert len(element) == self.N, f"Length of element ({len(element)}) must equal the atom number ({self.N})."
            names = np.array(element, dtype=object)
        self.update_data(self._frame.with_columns(element=names), True)

    def set_type_by_element(self, element_list):
        """column ``type`` = 1-based position of an atom's element in ``element_list`` (system.py:379-431)"""
        assert "element" in self.data.columns, "Data must contain element column."
        names = self.data["element"].to_numpy()
        order = {name: k for k, name in enumerate(element_list, start=1)}
        for name in np.unique(names):
            assert name in order, f"element_list must include element {name!r} (seen in data['element'])."
        present, codes = policy.label_codes(names)
        lut = np.array([order[name] for name in present], dtype=np.int32)
        self.update_data(self._frame.with_columns(type=lut[np.asarray(codes)]), True)

    def get_positions(self, reduced=False):
        """frame of x, y, z — or, reduced, of r_x, r_y, r_z = positions @ inverse box (system.py:433-477; like the reference the
        origin is not subtracted)"""
        if not reduced:
            return self.data.select("x", "y", "z")
        inv = self.box.inverse_box
        x, y, z = (self.data[c].to_numpy() for c in ("x", "y", "z"))
        return Frame({"r_x": x * inv[0, 0] + y * inv[1, 0] + z * inv[2, 0], "r_y": x * inv[0, 1] + y * inv[1, 1] + z * inv[2, 1],
                      "r_z": x * inv[0, 2] + y * inv[1, 2] + z * inv[2, 2]})

    def get_velocities(self):
        for name in ("vx", "vy", "vz"):
            assert name in self.data.columns
        return self.data.select("vx", "vy", "vz")

    def update_box(self, box, scale_pos=False):
        """a new box; ``scale_pos`` moves the atoms affinely with it (system.py:786-846: all three axes periodic)"""
        new_box = box if isinstance(box, Box) else Box(box)
        if scale_pos:
            assert sum(new_box.boundary) == 3, "only support all periodic boundary condition."
            m = np.linalg.solve(self.box.box, new_box.box)
            x, y, z = (self.data[c].to_numpy() - self.box.origin[k] for k, c in enumerate(("x", "y", "z")))
            self._frame = self._frame.with_columns(x=x * m[0, 0] + y * m[1, 0] + z * m[2, 0] + new_box.origin[0],
                                                   y=x * m[0, 1] + y * m[1, 1] + z * m[2, 1] + new_box.origin[1],
                                                   z=x * m[0, 2] + y * m[1, 2] + z * m[2, 2] + new_box.origin[2])
        self.box = new_box  # (the setter forgets the neighbor list)

    def delete_overlap(self, rc, max_neigh=None):
        """remove every atom that has a SURVIVING neighbour of smaller index closer than ``rc``; returns how many went
        (system.py:1414-1490: a sweep in index order).  The sweep's answer is the unique solution of
        gone[j] = any(not gone[i] for the neighbours i < j within rc); it is reached here by repeating that rule over all
        atoms at once until nothing changes (a chain of k overlapping atoms takes k rounds)."""
        rc = float(rc)
        assert rc > 0, "rc should be larger than 0."
        n = self.N
        if not (hasattr(self, "rc") and self.rc >= rc):
            self.build_neighbor(rc, max_neigh)
        rows, dist = as_numpy(self.verlet_list), as_numpy(self.distance_list)
        if "_enlarge_data" in self.__dict__:
            rows = np.where(rows >= 0, rows % n, -1)  # replica indices back to the atoms they copy
        rows, dist = rows[:n], dist[:n]
        own = np.arange(n)[:, None]
        smaller = (rows >= 0) & (rows < own) & (dist < rc)
        gone = np.zeros(n, dtype=bool)
        cols = np.where(smaller, rows, 0)
        while True:
            nxt = (smaller & ~gone[cols]).any(axis=1)
            if np.array_equal(nxt, gone):
                break
            gone = nxt
        removed = int(gone.sum())
        if removed:
            self.update_data(self.data.filter(~gone), reset_calculator=True, reset_neighbor=True)
        return removed

    def replicate(self, nx, ny, nz):
        self._frame, self.box = tool.replicate(self._frame, self.box, nx, ny, nz)

    # ------------------------------------------------------- neighbor lists
    def _remember(self, search, rows, distances, counts):
        """take over the list a search object built, and its replica if it made one (the replica always belongs to the
        list that is current: one left behind by an earlier search would be paired with rows it does not describe)"""
        self._forget(_REPLICA)
        for name in _REPLICA:
            if name in search.__dict__:
                setattr(self, name, getattr(search, name))
        self.verlet_list, self.distance_list, self.neighbor_number = rows, distances, counts
        self._sorted_columns = (id(rows), 0)  # (the list it speaks of, leading columns known to hold the nearest neighbours in order)

    def _sort_front(self, k):
        """the k nearest of every row to the front, nearest first — once: a row prefix that is already in order (the rows of a
        k-nearest search, or an earlier call) is not touched again"""
        which, done = self.__dict__.get("_sorted_columns", (None, 0))
        if which != id(self.verlet_list) or done < k:
            tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, k)
            self._sorted_columns = (id(self.verlet_list), k)

    @_on_twin
    def build_neighbor(self, rc, max_neigh=None, _label=False):
        # _label (internal): the fixed-cutoff common-neighbour labels of this cutoff in the same pass; returned, not stored
        search = Neighbor(rc, self.box, self.data, max_neigh, key=self.__dict__.get("_order_key"))
        search.compute(label=_label)
        self.rc = rc
        self._remember(search, search.verlet_list, search.distance_list, search.neighbor_number)
        # provenance of the CURRENT list: a cutoff list of exactly this reach, complete (an overflow of max_neigh raises).
        # `rc` alone does not say so — like the reference's, it survives build_nearest_neighbor (system.py:1256-1263)
        self._list_cutoff = float(rc)
        return search.pattern if _label else None

    @_on_twin
    def build_nearest_neighbor(self, k):
        """k nearest neighbours as the current list (no ``rc``; every count is k)"""
        search = NearestNeighbor(self.data, self.box, k)
        search.compute()
        # (the counts live where the rows live: a host array here was 40 MB over PCIe in front of every analysis that takes the list)
        self._remember(search, search.indices_py, search.distances_py, full((search.indices_py.shape[0],), k, np.int32))
        self._forget(("_list_cutoff",))  # the current list is no cutoff list any more (`rc` stays, as in the reference)
        self._sorted_columns = (id(self.verlet_list), k)

    def _require_cutoff_list(self, rc, max_neigh):
        """a cutoff list reaching at least rc: the remembered one if it does, a new one otherwise"""
        if not ("rc" in self.__dict__ and self.rc >= rc):
            self.build_neighbor(rc, max_neigh)

    def _deep_enough(self, k):
        return "neighbor_number" in self.__dict__ and self.neighbor_number.min() >= k

    def _nearest_prefix(self, k, cutoff_lists_only=False):
        """make the first k columns of the current list the k nearest neighbours, nearest first: by sorting the remembered
        list when every atom has k entries, by a k-nearest search otherwise"""
        if self._deep_enough(k) and (not cutoff_lists_only or "rc" in self.__dict__):
            self._sort_front(k)
        else:
            self.build_nearest_neighbor(k)

    def _safe_repeat(self, safe_L=15):
        return policy.axis_copies(self.box, safe_L)

    def _borrow_nearest(self, k):
        """rows to lend to an analysis that otherwise searches its k nearest neighbours itself: the remembered list,
        sorted, if it is deep enough — but never for a box so thin that the analysis would replicate it (the list of the
        unreplicated system would miss images)"""
        if policy.is_single(self._safe_repeat()) and self._deep_enough(k):
            self._sort_front(k)
            return self.verlet_list
        return None

    # ------------------------------------------------------------- analyses
    @_on_twin
    def cal_common_neighbor_analysis(self, rc=None, max_neigh=None):
        """column ``cna``: 0 other, 1 fcc, 2 hcp, 3 bcc, 4 ico; fixed cutoff ``rc`` or adaptive (None)"""
        rows = counts = None
        if rc is None:
            if "rc" in self.__dict__:  # (only a cutoff list is lent to the adaptive variant)
                rows = self._borrow_nearest(14)
        elif policy.is_single(self._safe_repeat()) and not ("rc" in self.__dict__ and self.rc >= rc):
            # no list that reaches: the reference builds one (it stays the system's list) and labels from it; here the labels
            # are made in the pass that builds the list (mdh_build_neighbor_fcna / _exact_fcna), bit for bit the same
            labels = self.build_neighbor(rc, max_neigh, _label=True)
            if labels is not None:
                self._store(cna=labels)
                return
            rows, counts = self.verlet_list, self.neighbor_number
        elif policy.is_single(self._safe_repeat()) and self.__dict__.get("_list_cutoff") == float(rc):
            # the reference lends nothing here and the analysis builds a list of its own with this very cutoff: the same
            # rows and counts as the CURRENT list when that is a cutoff list of exactly this reach, which is lent instead
            # (one neighbour build less per call).  A stale `rc` beside a k-nearest list lends nothing.
            rows, counts = self.verlet_list, self.neighbor_number
        cell, frame = self._get_compute_view()
        job = CommonNeighborAnalysis(frame, cell, rows, counts, rc)
        job.compute()
        self._store(cna=job.pattern)

    def cal_polyhedral_template_matching(self, structure="fcc-hcp-bcc", rmsd_threshold=0.1, return_ordering=False,
                                         return_rmsd=False, return_atomic_distance=False, return_orientation=False,
                                         identify_fcc_planar_faults=False, identify_esf=True):
        """column ``ptm`` and, on request, ``ordering``, ``rmsd``, ``interatomic_distance``, ``qx qy qz qw``, ``pft``"""
        twin = self._twin_for("cal_polyhedral_template_matching", (), {})
        if twin is not None:
            # the matching on the twin; the planar-fault sweep (its answer depends on the atom numbering: an index-ordered sweep,
            # identify_fcc_planar_faults.cpp:139-180) here, on the translated labels and template-ordered neighbours
            self._run_on_twin(twin, "cal_polyhedral_template_matching", (structure, rmsd_threshold, return_ordering, return_rmsd,
                                                                        return_atomic_distance, return_orientation, False, identify_esf), {})
            if identify_fcc_planar_faults:
                shell = np.ascontiguousarray(as_numpy(self.ptm_indices)[:, 1:13])
                faults = IdentifyFccPlanarFaults(np.array(self.data["ptm"].to_numpy(), np.int32), shell, identify_esf)
                faults.compute()
                self._store(pft=faults.fault_types)
            return
        rows = self._borrow_nearest(18)
        cell, frame = self._get_compute_view()
        job = PolyhedralTemplateMatching(structure, frame, cell, rmsd_threshold, rows)
        job.compute()
        table = job.output
        if isinstance(table, HArray):  # columns are cut out in HBM; nothing crosses PCIe before it is asked for
            take = table.column
        else:
            take = lambda col, dtype=None: table[:, col] if dtype is None else table[:, col].astype(dtype)
        found = {"ptm": take(0, np.int32)}
        for wanted, name, col in ((return_ordering, "ordering", 1), (return_rmsd, "rmsd", 2),
                                  (return_atomic_distance, "interatomic_distance", 3)):
            if wanted:
                found[name] = take(col)
        if return_orientation:  # stored x, y, z, w; the kernel's quaternion is w, x, y, z
            found.update(qx=take(5), qy=take(6), qz=take(7), qw=take(4))
        self.ptm_indices = job.ptm_indices
        if identify_fcc_planar_faults:
            shell = np.ascontiguousarray(as_numpy(job.ptm_indices)[:, 1:13])  # the 12 neighbours in template order
            faults = IdentifyFccPlanarFaults(np.array(as_numpy(found["ptm"]), np.int32), shell, identify_esf)
            faults.compute()
            found["pft"] = faults.fault_types
        self._store(**found)

    @_on_twin
    def cal_common_neighbor_parameter(self, rc, max_neigh=None):
        """column ``cnp``"""
        self._require_cutoff_list(rc, max_neigh)
        cell, frame = self._get_compute_view()
        job = CommonNeighborParameter(frame, cell, rc, self.verlet_list, self.distance_list, self.neighbor_number)
        job.compute()
        self._store(cnp=job.cnp)

    @_on_twin
    def cal_ackland_jones_analysis(self):
        """column ``aja``: 0 other, 1 fcc, 2 hcp, 3 bcc, 4 ico"""
        depth = 14
        if self.N < depth and int(np.sum(self.box.boundary)) == 0:
            self._store(aja=np.zeros(self.N, np.int32))
            return
        self._nearest_prefix(depth)
        cell, frame = self._get_compute_view()
        job = AcklandJonesAnalysis(frame, cell, self.verlet_list, self.distance_list)
        job.compute()
        self._store(aja=job.aja)

    @_on_twin
    def cal_structure_entropy(self, rc, sigma, use_local_density=False, average_rc=0.0, max_neigh=None):
        """column ``entropy`` (and ``entropy_ave`` when average_rc > 0)"""
        self._require_cutoff_list(rc, max_neigh)
        cell, _ = self._get_compute_view()
        job = StructureEntropy(cell, self.verlet_list, self.distance_list, self.neighbor_number, rc, sigma,
                               use_local_density, average_rc)
        job.compute()
        found = {"entropy": job.entropy}
        if average_rc > 0:
            found["entropy_ave"] = job.entropy_ave
        self._store(**found)

    @_on_twin
    def cal_atomic_temperature(self, rc, factor=1.0, max_neigh=None):
        """column ``atomic_temp`` (K); velocities are A/fs times ``factor``"""
        self._require_cutoff_list(rc, max_neigh)
        job = AtomicTemperature(self._get_compute_view()[1], self.verlet_list, self.distance_list, rc, factor)
        job.compute()
        self._store(atomic_temp=job.T)

    def cal_cluster_analysis(self, rc=5.0, max_neigh=None):
        """column ``cluster_id`` and attribute ``cluster_number``; ``rc`` one number or a dict per type pair"""
        per_pair = isinstance(rc, dict)
        if per_pair:
            reach = max(rc.values())
        elif isinstance(rc, (int, float, np.integer, np.floating)):
            reach = float(rc)
        else:
            raise TypeError("rc should be a positive number, or a dict like {'1-1':1.5, '1-2':1.3}")
        self._require_cutoff_list(reach, max_neigh)
        types = None
        if per_pair:
            if "type" not in self.data.columns:
                raise AssertionError("Must have type for multi rc cluster calculation.")
            # types of the atoms the list indexes, i.e. of the replica when there is one
            types = np.ascontiguousarray(self._get_compute_view()[1]["type"].to_numpy(), dtype=np.int32)
        job = ClusterAnalysis(rc, self.verlet_list, self.distance_list, self.neighbor_number, types)
        job.compute()
        self.cluster_number = job.cluster_number
        self._store(cluster_id=job.particleClusters)

    def build_voronoi_neighbor(self, a_face_area_threshold=-1.0, r_face_area_threshold=-1.0):
        """``voro_verlet_list``, ``voro_distance_list``, ``voro_face_area``, ``voro_neighbor_number``"""
        cells = Voronoi(self.box, self.data)
        lists = cells.get_neighbor(a_face_area_threshold, r_face_area_threshold)
        self.voro_verlet_list, self.voro_distance_list, self.voro_face_area, self.voro_neighbor_number = lists
        for name in _REPLICA:
            if name in cells.__dict__:
                setattr(self, name, getattr(cells, name))

    def cal_structure_factor(self, k_min, k_max, nbins, cal_partial=False, atomic_form_factors=False, mode="direct",
                             rc=None, nbin_rdf=200, window=False):
        """-> StructureFactor with ``k``, ``Sk``, ``Sk_partial``"""
        job = StructureFactor(self.data, self.box, k_min, k_max, nbins, cal_partial, atomic_form_factors, mode, rc,
                              nbin_rdf, window)
        job.compute()
        return job

    def cal_voronoi_volume(self):
        """columns ``volume``, ``neighbor_number`` (faces of the cell), ``cavity_radius``"""
        volume, faces, radius = Voronoi(self.box, self.data).get_volume()
        self.update_data(self.data.with_columns(volume=volume, neighbor_number=faces, cavity_radius=radius))

    @_on_twin
    def cal_centro_symmetry_parameter(self, N):
        """column ``csp`` from the N nearest neighbours (N even)"""
        if not (N > 0 and N % 2 == 0):
            raise AssertionError(f"N must be a positive even number: {N}.")
        if self.N <= N and int(np.sum(self.box.boundary)) == 0:
            self._store(csp=np.full(self.N, 10000, float))  # an open system with too few atoms: "as asymmetric as it gets"
            return
        self._nearest_prefix(N, cutoff_lists_only=True)
        cell, frame = self._get_compute_view()
        job = CentroSymmetryParameter(frame, cell, N, self.verlet_list)
        job.compute()
        self._store(csp=job.csp)

    def cal_identify_diamond_structure(self):
        """column ``ids``: 0 other, 1 cubic diamond (2, 3 its shells), 4 hexagonal diamond (5, 6 its shells)"""
        rows = self._borrow_nearest(4)
        cell, frame = self._get_compute_view()
        job = IdentifyDiamondStructure(frame, cell, rows)
        job.compute()
        self._store(ids=job.pattern)

    @_on_twin
    def cal_steinhardt_bond_orientation(self, llist, use_voronoi=False, nnn=0, rc=-1.0, average=False, use_weight=False,
                                        weight=None, wl=False, wlhat=False, a_face_area_threshold=-1,
                                        r_face_area_threshold=-1, identify_liquid=False, threshold=0.7, n_bond=7,
                                        max_neigh=None):
        """columns ``ql{l}`` (and ``wl{l}``, ``wlh{l}``, ``solidliquid``, ``nbond`` on request)"""
        if use_voronoi:
            self.build_voronoi_neighbor(a_face_area_threshold, r_face_area_threshold)
            lists = (self.voro_verlet_list, self.voro_distance_list, self.voro_neighbor_number)
            if use_weight and weight is None:
                weight = self.voro_face_area
        else:
            if nnn > 0:
                self._nearest_prefix(nnn)
            else:
                assert rc > 0, "At least use voronoi, or set positive nnn, or positive rc."
                self._require_cutoff_list(rc, max_neigh)
            lists = (self.verlet_list, self.distance_list, self.neighbor_number)
        cell, frame = self._get_compute_view()
        job = SteinhardtBondOrientation(cell, frame, np.asarray(llist, int), nnn, rc, average, use_voronoi, use_weight,
                                        weight, *lists, wl, wlhat, identify_liquid, threshold, n_bond)
        job.compute()
        values = job.qnarray
        take = values.column if isinstance(values, HArray) else (lambda col: values[:, col])
        if values.shape[1] == 1:
            found = {f"ql{llist[0]}": take(0)}
        else:
            names = [f"ql{l}" for l in llist]
            names += [f"wl{l}" for l in llist] if wl else []
            names += [f"wlh{l}" for l in llist] if wlhat else []
            found = {name: take(col) for col, name in enumerate(names)}
        if identify_liquid:
            found.update(solidliquid=job.solidliquid, nbond=job.nbond)
        self._store(**found)

    @_on_twin
    def cal_radial_distribution_function(self, rc, nbin=100, max_neigh=None, streaming=None):
        """-> RadialDistributionFunction (``r``, ``g_total``, ``g_partial``).  ``streaming=None`` decides by itself: a cutoff
        of a third of the thinnest periodic direction or more is counted straight from the positions, without a list"""
        cell, frame = self._get_compute_view()
        if streaming is None:
            spans = [t for t, periodic in zip(cell.get_thickness(), cell.boundary) if periodic]
            streaming = rc >= (min(spans) if spans else float("inf")) / 3.0
        if streaming:
            copies = self.box.check_small_box(rc)
            if not policy.is_single(copies):
                frame, cell = tool.replicate(frame, cell, *copies)
            job = RadialDistributionFunction(rc, nbin, cell, type_list=policy.species_of(frame), streaming=True,
                                             x=frame["x"], y=frame["y"], z=frame["z"])
        else:
            self._require_cutoff_list(rc, max_neigh)
            cell, frame = self._get_compute_view()
            job = RadialDistributionFunction(rc, nbin, cell, verlet_list=self.verlet_list, distance_list=self.distance_list,
                                             neighbor_number=self.neighbor_number, type_list=policy.species_of(frame))
        job.compute()
        return job

    @_on_twin
    def cal_warren_cowley_parameter(self, rc, max_neigh=None):
        """-> WarrenCowleyParameter (``WCP`` matrix)"""
        self._require_cutoff_list(rc, max_neigh)
        job = WarrenCowleyParameter(self.verlet_list, self.neighbor_number, self._get_compute_view()[1])
        job.compute()
        return job

    @_on_twin
    def average_by_neighbor(self, average_rc, property_name, include_self=True, output_name=None, max_neigh=None):
        """column ``<property>_ave`` (or ``output_name``): neighbourhood mean of a column within ``average_rc``"""
        self._require_cutoff_list(average_rc, max_neigh)
        averaged = tool.average_by_neighbor(average_rc, self._get_compute_view()[1], property_name, self.verlet_list, self.distance_list,
                                            self.neighbor_number, include_self, output_name)
        name = f"{property_name}_ave" if output_name is None else output_name
        self._store(**{name: averaged[name].to_numpy()})
