"""The one door between the host layer and the C ABI.

Every analysis class reaches libmdapy_amd.so through the attributes of this module — one shim module per nanobind
extension of the reference (``mdapy._neighbor`` ... ``mdapy._repeat_cell``, CMakeLists.txt:71-100; same function names and
argument order, implemented by ctypes calls into include/mdapy_amd.h).  Keeping the door in one place is what lets the
CPU test-suite swap the shims for the oracle without a backend switch inside the package."""
from . import _aja as aja
from . import _atomtemp as atomtemp
from . import _cluster as cluster
from . import _cna as cna
from . import _cnp as cnp
from . import _csp as csp
from . import _fast_knn as fast_knn
from . import _fccpft as fccpft
from . import _neighbor as neighbor
from . import _order as order
from . import _polycrystal as polycrystal
from . import _ptm as ptm
from . import _rdf as rdf
from . import _repeat_cell as repeat_cell
from . import _sbo as sbo
from . import _sfc as sfc
from . import _structure_entropy as structure_entropy
from . import _voronoi as voronoi
from . import _wcp as wcp

NAMES = ("aja", "atomtemp", "cluster", "cna", "cnp", "csp", "fast_knn", "fccpft", "neighbor", "order", "polycrystal", "ptm", "rdf",
         "repeat_cell", "sbo", "sfc", "structure_entropy", "voronoi", "wcp")
