"""mdapy_amd — MI355X-native (gfx950, hand-written HIP) implementation of mdapy's
neighbor-list + per-atom structural-analysis hot path, behind mdapy's own
``System`` / ``cal_*`` / ``_module.function`` API.  See DESIGN.md."""
__version__ = "0.1.0"

from .box import Box
from .frame import Frame
from .system import System
from .neighbor import Neighbor
from .knn import NearestNeighbor
from .common_neighbor_analysis import CommonNeighborAnalysis
from .centro_symmetry_parameter import CentroSymmetryParameter
from .identify_diamond_structure import IdentifyDiamondStructure
from .steinhardt_bond_orientation import SteinhardtBondOrientation
from .polyhedral_template_matching import PolyhedralTemplateMatching
from .radial_distribution_function import RadialDistributionFunction
from .warren_cowley_parameter import WarrenCowleyParameter
from .build_lattice import build_crystal
from .create_polycrystal import CreatePolycrystal
from .parallel import get_num_threads

__all__ = [
    "Box", "Frame", "System", "Neighbor", "NearestNeighbor", "CommonNeighborAnalysis", "CentroSymmetryParameter",
    "IdentifyDiamondStructure", "SteinhardtBondOrientation", "PolyhedralTemplateMatching", "RadialDistributionFunction", "WarrenCowleyParameter",
    "build_crystal", "CreatePolycrystal", "get_num_threads",
]
