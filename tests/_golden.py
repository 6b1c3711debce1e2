"""Golden data of the reference test-suite (tests/golden: plain data files copied from
/root/reference/tests/fixtures and tests/input_files; see tests/golden/README.md)."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).parent / "golden"
SA_DIR = GOLDEN / "structure_analysis"


def fixtures_with(key):
    out = []
    for p in sorted(SA_DIR.glob("*.npz")):
        with np.load(p) as d:
            if key in d.files:
                out.append(p)
    return out


def ids_of(paths):
    return [p.stem for p in paths]


def misc(name):
    return np.load(GOLDEN / "misc" / f"{name}.npz")


def input_path(name):
    return str(GOLDEN / "input_files" / name)


def system_from_fixture(d):
    import mdapy_amd as mp

    return mp.System(pos=d["pos"], box=mp.Box(d["box"], boundary=list(d["boundary"])))
