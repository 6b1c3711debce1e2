"""``System``: the user-facing object of the hot path.

Mirrors the part of ``mdapy.System`` (src/mdapy/system.py) that sits above the
neighbor list and the per-atom structural analyses: construction from
(pos, box) / (data, box) / a file, the cached neighbor attributes
(``verlet_list``, ``distance_list``, ``neighbor_number``, ``rc``,
``_enlarge_data``, ``_enlarge_box`` — system.py:1155-1166), their invalidation
rules (:232-245, :748-763), and ``build_neighbor`` / ``build_nearest_neighbor`` /
``cal_*`` with the reference's list-reuse and replication policy.  Results land
as columns of ``system.data`` or as returned objects, like in the reference.

The neighbor arrays stay in HBM between calls (:class:`mdapy_amd.devarray.HArray`);
they convert to numpy on first host access.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np

from . import tool_function as tool
from .ackland_jones_analysis import AcklandJonesAnalysis
from .atomic_temperature import AtomicTemperature
from .box import Box
from .centro_symmetry_parameter import CentroSymmetryParameter
from .cluster_analysis import ClusterAnalysis
from .common_neighbor_analysis import CommonNeighborAnalysis
from .common_neighbor_parameter import CommonNeighborParameter
from .devarray import as_numpy
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

_NEIGH_ATTRS = ("verlet_list", "neighbor_number", "distance_list", "rc", "_enlarge_box", "_enlarge_data")


class System:
    def __init__(self, filename: Optional[str] = None, data=None, pos: Optional[np.ndarray] = None, box=None,
                 format: Optional[str] = None, global_info: Optional[Dict[str, Any]] = None):
        self.__global_info: Dict[str, Any] = {}
        if isinstance(filename, str):
            from .load_save import read_file

            self.__data, self.box, self.__global_info = read_file(filename, format)
        elif data is not None and box is not None:
            frame = Frame.from_any(data)
            for c in ("x", "y", "z"):
                assert c in frame.columns, f"data must contain column {c!r}."
            self.__data, self.box = frame, box
        elif pos is not None and box is not None:
            pos = np.asarray(pos, dtype=np.float64)
            assert pos.ndim == 2 and pos.shape[1] == 3, "pos must have shape (N, 3)."
            self.__data = Frame({"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]})
            self.box = box
        else:
            raise RuntimeError("One must at least provide filename or [data, box] or [pos, box].")
        if not len(self.__global_info) and global_info is not None:
            self.__global_info = dict(global_info)

    # ------------------------------------------------------------------ state
    @property
    def box(self) -> Box:
        return self.__box

    @box.setter
    def box(self, value):
        """Assigning a box keeps Cartesian coordinates and drops everything that depends on it (system.py:232-245)."""
        self.__box = value if isinstance(value, Box) else Box(value)
        for attr in _NEIGH_ATTRS:
            if hasattr(self, attr):
                delattr(self, attr)

    @property
    def data(self) -> Frame:
        return self.__data

    @property
    def global_info(self) -> Dict[str, Any]:
        return self.__global_info

    @property
    def N(self) -> int:
        return self.__data.shape[0]

    def __repr__(self) -> str:
        return f"Atom Number: {self.N}\n{self.box}\nParticle Information:\n{self.data}"

    def update_data(self, data, reset_calculator: bool = False, reset_neighbor: bool = False) -> None:
        """Replace the per-atom frame; ``reset_neighbor`` drops the cached lists (system.py:686-763)."""
        self.__data = Frame.from_any(data)
        if reset_neighbor:
            for attr in _NEIGH_ATTRS:
                if hasattr(self, attr):
                    delattr(self, attr)

    def _get_compute_view(self) -> Tuple[Box, Frame]:
        """(box, data) the cached neighbor indices refer to (system.py:765-784)"""
        if hasattr(self, "_enlarge_data"):
            return self._enlarge_box, self._enlarge_data
        return self.box, self.data

    def wrap_pos(self) -> None:
        self.update_data(tool.wrap_pos(self.__data, self.box), reset_neighbor=True)

    def replicate(self, nx: int, ny: int, nz: int) -> None:
        data, box = tool.replicate(self.__data, self.box, nx, ny, nz)
        self.__data = data
        self.box = box

    # ------------------------------------------------------------ neighbor lists
    def build_neighbor(self, rc: float, max_neigh: Optional[int] = None) -> None:
        """system.py:1108-1166"""
        neigh = Neighbor(rc, self.box, self.data, max_neigh)
        neigh.compute()
        self.rc = rc
        for attr in ("_enlarge_box", "_enlarge_data"):  # see build_nearest_neighbor: the view belongs to the current list
            if hasattr(self, attr):
                delattr(self, attr)
        if hasattr(neigh, "_enlarge_box"):
            self._enlarge_box = neigh._enlarge_box
        if hasattr(neigh, "_enlarge_data"):
            self._enlarge_data = neigh._enlarge_data
        self.verlet_list, self.distance_list, self.neighbor_number = (
            neigh.verlet_list, neigh.distance_list, neigh.neighbor_number)

    def build_nearest_neighbor(self, k: int) -> None:
        """system.py:1226-1263 (sets no ``rc``; neighbor_number = k everywhere)"""
        kdt = NearestNeighbor(self.data, self.box, k)
        kdt.compute()
        # The reference only ever SETS these (system.py:1257-1260): a replicated view left behind by an earlier cutoff
        # build would then be paired with rows of the unreplicated system (out-of-bounds reads downstream).  The view
        # always belongs to the list that is current.
        for attr in ("_enlarge_box", "_enlarge_data"):
            if hasattr(self, attr):
                delattr(self, attr)
        if hasattr(kdt, "_enlarge_box"):
            self._enlarge_box = kdt._enlarge_box
        if hasattr(kdt, "_enlarge_data"):
            self._enlarge_data = kdt._enlarge_data
        self.verlet_list, self.distance_list = kdt.indices_py, kdt.distances_py
        self.neighbor_number = np.full(self.verlet_list.shape[0], k, np.int32)

    def _safe_repeat(self, safe_L: float = 15) -> np.ndarray:
        repeat = np.ceil(safe_L / self.box.get_thickness()).astype(int)
        for i in range(3):
            if self.box.boundary[i] == 0:
                repeat[i] = 1
        return repeat

    # ------------------------------------------------------------------ analyses
    def cal_common_neighbor_analysis(self, rc: Optional[float] = None, max_neigh: Optional[int] = None):
        """column ``cna`` (system.py:2005-2064)"""
        verlet_list = neighbor_number = None
        if sum(self._safe_repeat()) == 3:
            if hasattr(self, "rc"):
                if rc is None:
                    if self.neighbor_number.min() >= 14:
                        tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, 14)
                        verlet_list = self.verlet_list
                elif self.rc < rc:
                    self.build_neighbor(rc, max_neigh)
                    verlet_list, neighbor_number = self.verlet_list, self.neighbor_number
            elif rc is not None:
                self.build_neighbor(rc, max_neigh)
                verlet_list, neighbor_number = self.verlet_list, self.neighbor_number
        box, data = self._get_compute_view()
        cna = CommonNeighborAnalysis(data, box, verlet_list, neighbor_number, rc)
        cna.compute()
        self.update_data(self.__data.with_columns(cna=as_numpy(cna.pattern)[: self.N]))

    def cal_polyhedral_template_matching(self, structure="fcc-hcp-bcc", rmsd_threshold=0.1, return_ordering=False,
                                         return_rmsd=False, return_atomic_distance=False, return_orientation=False,
                                         identify_fcc_planar_faults=False, identify_esf=True):
        """column ``ptm`` (+ ``ordering``, ``rmsd``, ``interatomic_distance``, ``qx,qy,qz,qw``, ``pft``)
        (system.py:1863-1970)"""
        verlet_list = None
        if sum(self._safe_repeat()) == 3:
            if hasattr(self, "neighbor_number"):
                if self.neighbor_number.min() >= 18:
                    tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, 18)
                    verlet_list = self.verlet_list
        box, data = self._get_compute_view()
        ptm = PolyhedralTemplateMatching(structure, data, box, rmsd_threshold, verlet_list)
        ptm.compute()
        output = as_numpy(ptm.output)[: self.N]
        new = {"ptm": output[:, 0].astype(np.int32)}
        if return_ordering:
            new["ordering"] = output[:, 1]
        if return_rmsd:
            new["rmsd"] = output[:, 2]
        if return_atomic_distance:
            new["interatomic_distance"] = output[:, 3]
        if return_orientation:
            new.update(qx=output[:, 5], qy=output[:, 6], qz=output[:, 7], qw=output[:, 4])
        self.ptm_indices = ptm.ptm_indices
        if identify_fcc_planar_faults:  # system.py:1963-1968
            structure_types = np.array(as_numpy(ptm.output)[:, 0], np.int32)
            ptm12 = np.ascontiguousarray(as_numpy(ptm.ptm_indices)[:, 1:13])
            ifpt = IdentifyFccPlanarFaults(structure_types, ptm12, identify_esf)
            ifpt.compute()
            new["pft"] = ifpt.fault_types[: self.N]
        self.update_data(self.__data.with_columns(**new))

    def cal_common_neighbor_parameter(self, rc: float, max_neigh: Optional[int] = None) -> None:
        """column ``cnp`` (system.py:1572-1603)"""
        has_neigh = hasattr(self, "rc") and self.rc >= rc
        if not has_neigh:
            self.build_neighbor(rc, max_neigh)
        box, data = self._get_compute_view()
        cnp = CommonNeighborParameter(data, box, rc, self.verlet_list, self.distance_list, self.neighbor_number)
        cnp.compute()
        self.update_data(self.__data.with_columns(cnp=as_numpy(cnp.cnp)[: self.N]))

    def cal_ackland_jones_analysis(self) -> None:
        """column ``aja``: 0 other, 1 fcc, 2 hcp, 3 bcc, 4 ico (system.py:1605-1636)"""
        n_neigh = 14
        if self.data.shape[0] < n_neigh and sum(self.box.boundary) == 0:
            self.update_data(self.__data.with_columns(aja=np.zeros(self.N, np.int32)))
            return
        if hasattr(self, "neighbor_number") and self.neighbor_number.min() >= n_neigh:
            tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, n_neigh)
        else:
            self.build_nearest_neighbor(n_neigh)
        box, data = self._get_compute_view()
        aja = AcklandJonesAnalysis(data, box, self.verlet_list, self.distance_list)
        aja.compute()
        self.update_data(self.__data.with_columns(aja=as_numpy(aja.aja)[: self.N]))

    def cal_structure_entropy(self, rc: float, sigma: float, use_local_density: bool = False, average_rc: float = 0.0,
                              max_neigh: Optional[int] = None) -> None:
        """columns ``entropy`` (+ ``entropy_ave`` when average_rc > 0) (system.py:2481-2542)"""
        if hasattr(self, "rc"):
            if self.rc < rc:
                self.build_neighbor(rc, max_neigh)
        else:
            self.build_neighbor(rc, max_neigh)
        box, _ = self._get_compute_view()
        se = StructureEntropy(box, self.verlet_list, self.distance_list, self.neighbor_number, rc, sigma, use_local_density,
                              average_rc)
        se.compute()
        data = self.data.with_columns(entropy=as_numpy(se.entropy)[: self.N])
        if average_rc > 0:
            data = data.with_columns(entropy_ave=as_numpy(se.entropy_ave)[: self.N])
        self.update_data(data)

    def cal_atomic_temperature(self, rc: float, factor: float = 1.0, max_neigh: Optional[int] = None) -> None:
        """column ``atomic_temp`` in K; velocities in A/fs x factor (system.py:1678-1714)"""
        has_neigh = hasattr(self, "rc") and self.rc >= rc
        if not has_neigh:
            self.build_neighbor(rc, max_neigh)
        _, data = self._get_compute_view()
        at = AtomicTemperature(data, self.verlet_list, self.distance_list, rc, factor)
        at.compute()
        self.update_data(self.data.with_columns(atomic_temp=as_numpy(at.T)[: self.N]))

    def cal_cluster_analysis(self, rc=5.0, max_neigh: Optional[int] = None) -> None:
        """column ``cluster_id`` (system.py:2416-2479)"""
        if isinstance(rc, (int, float, np.integer, np.floating)):
            max_rc = float(rc)
        elif isinstance(rc, dict):
            max_rc = max(rc.values())
        else:
            raise TypeError("rc should be a positive number, or a dict like {'1-1':1.5, '1-2':1.3}")
        if hasattr(self, "rc"):
            if self.rc < max_rc:
                self.build_neighbor(max_rc, max_neigh)
        else:
            self.build_neighbor(max_rc, max_neigh)
        type_list = None
        if isinstance(rc, dict):
            assert "type" in self.data.columns, "Must have type for multi rc cluster calculation."
            # types of the atoms the list indexes: the replicated view when the box was small (the reference passes the
            # unreplicated column, system.py:2472, which its filter then indexes out of bounds)
            type_list = np.ascontiguousarray(self._get_compute_view()[1]["type"].to_numpy(), dtype=np.int32)
        ca = ClusterAnalysis(rc, self.verlet_list, self.distance_list, self.neighbor_number, type_list)
        ca.compute()
        self.cluster_number = ca.cluster_number
        self.update_data(self.data.with_columns(cluster_id=as_numpy(ca.particleClusters)[: self.N]))

    def build_voronoi_neighbor(self, a_face_area_threshold: float = -1.0, r_face_area_threshold: float = -1.0) -> None:
        """``voro_verlet_list / voro_distance_list / voro_face_area / voro_neighbor_number`` (system.py:1168-1230)"""
        vor = Voronoi(self.box, self.data)
        (self.voro_verlet_list, self.voro_distance_list, self.voro_face_area,
         self.voro_neighbor_number) = vor.get_neighbor(a_face_area_threshold, r_face_area_threshold)
        if hasattr(vor, "_enlarge_box"):
            self._enlarge_box = vor._enlarge_box
        if hasattr(vor, "_enlarge_data"):
            self._enlarge_data = vor._enlarge_data

    def cal_structure_factor(self, k_min: float, k_max: float, nbins: int, cal_partial: bool = False,
                             atomic_form_factors: bool = False, mode: str = "debye", rc: Optional[float] = None,
                             nbin_rdf: int = 200, window: bool = False) -> StructureFactor:
        """-> StructureFactor with ``k``, ``Sk``, ``Sk_partial`` (system.py: cal_structure_factor)"""
        sf = StructureFactor(self.data, self.box, k_min, k_max, nbins, cal_partial, atomic_form_factors, mode, rc, nbin_rdf, window)
        sf.compute()
        return sf

    def cal_voronoi_volume(self) -> None:
        """columns ``volume``, ``neighbor_number`` (faces), ``cavity_radius`` (system.py:2544-2573)"""
        vor = Voronoi(self.box, self.data)
        volume, neighbor_number, cavity_radius = vor.get_volume()
        self.update_data(self.data.with_columns(volume=volume, neighbor_number=neighbor_number, cavity_radius=cavity_radius))

    def cal_centro_symmetry_parameter(self, N: int):
        """column ``csp`` (system.py:1972-2003)"""
        assert N % 2 == 0 and N > 0, f"N must be a positive even number: {N}."
        if self.N <= N and sum(self.box.boundary) == 0:
            res = np.full(self.N, 10000, float)
        else:
            has_verlet = False
            if hasattr(self, "neighbor_number"):
                if self.neighbor_number.min() >= N and hasattr(self, "rc"):
                    tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, N)
                    has_verlet = True
            if not has_verlet:
                self.build_nearest_neighbor(N)
            box, data = self._get_compute_view()
            csp = CentroSymmetryParameter(data, box, N, self.verlet_list)
            csp.compute()
            res = as_numpy(csp.csp)[: self.N]
        self.update_data(self.data.with_columns(csp=res))

    def cal_identify_diamond_structure(self) -> None:
        """column ``ids`` (system.py:1493-1529)"""
        verlet_list = None
        if sum(self._safe_repeat()) == 3:
            if hasattr(self, "neighbor_number"):
                if self.neighbor_number.min() >= 4:
                    tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, 4)
                    verlet_list = self.verlet_list
        box, data = self._get_compute_view()
        ids = IdentifyDiamondStructure(data, box, verlet_list)
        ids.compute()
        self.update_data(self.__data.with_columns(ids=as_numpy(ids.pattern)[: self.N]))

    def cal_steinhardt_bond_orientation(self, llist, use_voronoi: bool = False, nnn: int = 0, rc: float = -1.0,
                                        average: bool = False, use_weight: bool = False, weight=None,
                                        wl: bool = False, wlhat: bool = False, a_face_area_threshold: float = -1,
                                        r_face_area_threshold: float = -1, identify_liquid: bool = False,
                                        threshold: float = 0.7, n_bond: int = 7,
                                        max_neigh: Optional[int] = None) -> None:
        """columns ``ql{l}`` (+ ``wl{l}``, ``wlh{l}``, ``solidliquid``, ``nbond``) (system.py:1716-1861)"""
        v_list = d_list = n_list = None
        if use_voronoi:  # system.py:1781-1789
            self.build_voronoi_neighbor(a_face_area_threshold, r_face_area_threshold)
            v_list, d_list, n_list = self.voro_verlet_list, self.voro_distance_list, self.voro_neighbor_number
            if use_weight and weight is None:
                weight = self.voro_face_area
        elif nnn > 0:
            has_sort_neigh = False
            if hasattr(self, "neighbor_number"):
                if self.neighbor_number.min() >= nnn:
                    tool.sort_neighbor(self.verlet_list, self.distance_list, self.neighbor_number, nnn)
                    has_sort_neigh = True
            if not has_sort_neigh:
                self.build_nearest_neighbor(nnn)
        else:
            assert rc > 0, "At least use voronoi, or set positive nnn, or positive rc."
            if hasattr(self, "rc"):
                if self.rc < rc:
                    self.build_neighbor(rc, max_neigh)
            else:
                self.build_neighbor(rc, max_neigh)
        box, data = self._get_compute_view()
        if use_voronoi and hasattr(self, "_enlarge_data"):
            box, data = self._enlarge_box, self._enlarge_data
        if v_list is None:
            v_list, d_list, n_list = self.verlet_list, self.distance_list, self.neighbor_number
        SBO = SteinhardtBondOrientation(box, data, np.asarray(llist, int), nnn, rc, average, use_voronoi, use_weight,
                                        weight, v_list, d_list, n_list, wl, wlhat, identify_liquid, threshold, n_bond)
        SBO.compute()
        qn = as_numpy(SBO.qnarray)
        new = {}
        if qn.shape[1] > 1:
            columns = [f"ql{i}" for i in llist]
            if wl:
                columns.extend(f"wl{i}" for i in llist)
            if wlhat:
                columns.extend(f"wlh{i}" for i in llist)
            for i, name in enumerate(columns):
                new[name] = qn[: self.N, i]
        else:
            new[f"ql{llist[0]}"] = qn.flatten()[: self.N]
        if identify_liquid:
            new["solidliquid"] = as_numpy(SBO.solidliquid)[: self.N]
            new["nbond"] = as_numpy(SBO.nbond)[: self.N]
        self.update_data(self.data.with_columns(**new))

    def cal_radial_distribution_function(self, rc: float, nbin: int = 100, max_neigh: Optional[int] = None,
                                         streaming: Optional[bool] = None) -> RadialDistributionFunction:
        """system.py:2235-2361"""
        box, data = self._get_compute_view()
        if streaming is None:  # :2279-2288
            thickness = box.get_thickness()
            periodic = [thickness[i] for i in range(3) if box.boundary[i]]
            min_thick = min(periodic) if periodic else float("inf")
            streaming = rc >= min_thick / 3.0

        def _species_labels(view):
            if "element" in view.columns:
                return view["element"].to_numpy()
            if "type" in view.columns:
                return view["type"].to_numpy()
            return np.zeros(view.shape[0], np.int32)

        if streaming:
            repeat = self.box.check_small_box(rc)
            if sum(repeat) != 3:
                rep_data, rep_box = tool.replicate(data, box, *repeat)
                rdf = RadialDistributionFunction(rc, nbin, rep_box, type_list=_species_labels(rep_data),
                                                 streaming=True, x=rep_data["x"], y=rep_data["y"], z=rep_data["z"])
            else:
                rdf = RadialDistributionFunction(rc, nbin, box, type_list=_species_labels(data), streaming=True,
                                                 x=data["x"], y=data["y"], z=data["z"])
        else:
            if not (hasattr(self, "rc") and self.rc >= rc):
                self.build_neighbor(rc, max_neigh)
            box, data = self._get_compute_view()
            rdf = RadialDistributionFunction(rc, nbin, box, verlet_list=self.verlet_list,
                                             distance_list=self.distance_list, neighbor_number=self.neighbor_number,
                                             type_list=_species_labels(data))
        rdf.compute()
        return rdf

    def cal_warren_cowley_parameter(self, rc: float, max_neigh: Optional[int] = None) -> WarrenCowleyParameter:
        """system.py:1638-1676"""
        if not (hasattr(self, "rc") and self.rc >= rc):
            self.build_neighbor(rc, max_neigh)
        _, data = self._get_compute_view()
        wcp = WarrenCowleyParameter(self.verlet_list, self.neighbor_number, data)
        wcp.compute()
        return wcp

    def average_by_neighbor(self, average_rc: float, property_name: str, include_self: bool = True,
                            output_name: Optional[str] = None, max_neigh: Optional[int] = None) -> None:
        """system.py:2363-2414"""
        if not (hasattr(self, "rc") and self.rc >= average_rc):
            self.build_neighbor(average_rc, max_neigh)
        _, data = self._get_compute_view()
        out = tool.average_by_neighbor(average_rc, data, self.verlet_list, self.distance_list, self.neighbor_number,
                                       property_name, include_self, output_name)
        name = output_name if output_name is not None else f"{property_name}_ave"
        self.update_data(self.data.with_columns(**{name: out[name].to_numpy()[: self.N]}))
