"""Minimal readers for the two text formats the hot-path tests use (extended XYZ
and LAMMPS dump).  The reference's I/O layer (src/mdapy/load_save.py, 2 kLoC) is
out of scope; these readers exist so that ``System(filename)``
works on the reference's own sample files."""
from __future__ import annotations

import gzip
import re
import shlex
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .box import Box
from .frame import Frame


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def _table(path, skiprows: int, nrows: int, names, kinds) -> Optional[Dict[str, np.ndarray]]:
    """The atom table of a text file through pandas' C tokenizer (exact ``float_precision="round_trip"`` conversion, so the
    values are those of Python's ``float()``): 10 M rows in seconds instead of a per-line Python loop.  ``kinds``: "f" / "i" /
    "s" per column.  None when pandas is not installed or refuses the table (the line-by-line parser then takes over)."""
    try:
        import pandas as pd
    except ImportError:
        return None
    dtypes = {n: {"f": np.float64, "i": np.int64, "s": str}[k] for n, k in zip(names, kinds)}
    try:
        df = pd.read_csv(path, sep=r"\s+", header=None, names=list(names), skiprows=skiprows, nrows=nrows, dtype=dtypes,
                         engine="c", float_precision="round_trip", na_filter=False, skipinitialspace=True)
    except Exception:
        return None
    if len(df) != nrows:
        return None
    return {n: (df[n].to_numpy(dtype=object) if k == "s" else np.ascontiguousarray(df[n].to_numpy())) for n, k in zip(names, kinds)}


def read_xyz(path) -> Tuple[Frame, Box, Dict[str, Any]]:
    with _open(path) as f:
        n = int(f.readline().split()[0])
        header = f.readline()
    kv = dict(re.findall(r'(\w+)=("[^"]*"|\S+)', header))
    kv = {k: v.strip('"') for k, v in kv.items()}
    if "Lattice" not in kv:
        raise ValueError(f"{path}: extended-XYZ header without Lattice=")
    box = np.array([float(v) for v in kv["Lattice"].split()], dtype=np.float64).reshape(3, 3)
    origin = np.array([float(v) for v in kv.get("Origin", "0 0 0").split()], dtype=np.float64)
    pbc = kv.get("pbc")
    boundary = [1, 1, 1] if pbc is None else [1 if t.upper().startswith("T") or t == "1" else 0 for t in pbc.split()]
    props = kv.get("Properties", "species:S:1:pos:R:3").split(":")
    cols: Dict[str, np.ndarray] = {}
    flat_names, flat_kinds = [], []
    for name, kind, cnt in zip(props[0::3], props[1::3], props[2::3]):
        for q in range(int(cnt)):
            flat_names.append(f"{name}#{q}")
            flat_kinds.append({"R": "f", "I": "i"}.get(kind, "s"))
    tab = _table(path, 2, n, flat_names, flat_kinds)
    if tab is not None:
        for name, kind, cnt in zip(props[0::3], props[1::3], props[2::3]):
            cnt = int(cnt)
            if name == "pos":
                cols["x"], cols["y"], cols["z"] = tab["pos#0"], tab["pos#1"], tab["pos#2"]
            elif name == "species":
                cols["element"] = tab["species#0"]
            elif cnt == 1:
                cols[name] = tab[f"{name}#0"]
            else:
                for q in range(cnt):
                    cols[f"{name}_{q}"] = tab[f"{name}#{q}"]
        ordered = {k: cols[k] for k in ("x", "y", "z")}
        ordered.update({k: v for k, v in cols.items() if k not in ordered})
        return Frame(ordered), Box(box, boundary, origin), {}
    with _open(path) as f:
        f.readline(); f.readline()
        rows = [f.readline().split() for _ in range(n)]
    c = 0
    for name, kind, cnt in zip(props[0::3], props[1::3], props[2::3]):
        cnt = int(cnt)
        vals = [r[c:c + cnt] for r in rows]
        c += cnt
        if name == "pos":
            arr = np.array(vals, dtype=np.float64)
            cols["x"], cols["y"], cols["z"] = arr[:, 0], arr[:, 1], arr[:, 2]
        elif name == "species":
            cols["element"] = np.array([v[0] for v in vals], dtype=object)
        else:
            dt = {"R": np.float64, "I": np.int64, "S": object}.get(kind, object)
            arr = np.array(vals, dtype=dt)
            if cnt == 1:
                cols[name] = arr[:, 0]
            else:
                for q in range(cnt):
                    cols[f"{name}_{q}"] = arr[:, q]
    ordered = {k: cols[k] for k in ("x", "y", "z")}
    ordered.update({k: v for k, v in cols.items() if k not in ordered})
    return Frame(ordered), Box(box, boundary, origin), {}


def read_dump(path) -> Tuple[Frame, Box, Dict[str, Any]]:
    with _open(path) as f:
        head = [f.readline() for _ in range(64)]  # the header of a LAMMPS dump frame is 9 lines
    at = next((q for q, ln in enumerate(head) if ln.startswith("ITEM: ATOMS")), None)
    lines = None
    if at is not None:
        lines = [ln.rstrip("\n") for ln in head[: at + 1]]
    else:
        with _open(path) as f:
            lines = f.read().splitlines()
    info: Dict[str, Any] = {}
    i = 0
    n = 0
    box = boundary = origin = None
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("ITEM: TIMESTEP"):
            info["timestep"] = int(float(lines[i + 1]))
            i += 2
        elif ln.startswith("ITEM: NUMBER OF ATOMS"):
            n = int(lines[i + 1])
            i += 2
        elif ln.startswith("ITEM: BOX BOUNDS"):
            toks = ln.split()[3:]
            tri = "xy" in toks
            flags = [t for t in toks if t not in ("xy", "xz", "yz")]
            boundary = [1 if t == "pp" else 0 for t in flags] if flags else [1, 1, 1]
            vals = [[float(v) for v in lines[i + 1 + d].split()] for d in range(3)]
            if tri:
                (xlo_b, xhi_b, xy), (ylo_b, yhi_b, xz), (zlo, zhi, yz) = vals
                xlo = xlo_b - min(0.0, xy, xz, xy + xz)
                xhi = xhi_b - max(0.0, xy, xz, xy + xz)
                ylo = ylo_b - min(0.0, yz)
                yhi = yhi_b - max(0.0, yz)
                box = np.array([[xhi - xlo, 0, 0], [xy, yhi - ylo, 0], [xz, yz, zhi - zlo]], dtype=np.float64)
            else:
                (xlo, xhi), (ylo, yhi), (zlo, zhi) = [v[:2] for v in vals]
                box = np.diag([xhi - xlo, yhi - ylo, zhi - zlo]).astype(np.float64)
            origin = np.array([xlo, ylo, zlo], dtype=np.float64)
            i += 4
        elif ln.startswith("ITEM: ATOMS"):
            names = ln.split()[2:]
            kinds = ["i" if nm in ("id", "type") else ("s" if nm == "element" else "f") for nm in names]
            tab = _table(path, i + 1, n, names, kinds)
            if tab is not None:
                cols = dict(tab)
                if "type" in cols:
                    cols["type"] = cols["type"].astype(np.int32)
                ordered = {k: cols[k] for k in ("x", "y", "z")}
                ordered.update({k: v for k, v in cols.items() if k not in ordered})
                return Frame(ordered), Box(box, boundary, origin), info
            if len(lines) < i + 1 + n:  # only the header was read so far
                with _open(path) as f:
                    lines = f.read().splitlines()
            body = np.array([l.split() for l in lines[i + 1:i + 1 + n]], dtype=object)
            cols: Dict[str, np.ndarray] = {}
            for q, name in enumerate(names):
                col = body[:, q]
                if name in ("id", "type"):
                    cols[name] = col.astype(np.int64).astype(np.int32 if name == "type" else np.int64)
                elif name == "element":
                    cols[name] = col
                else:
                    cols[name] = col.astype(np.float64)
            # rows stay in file order, as in the reference (load_save.py:66-200 keeps the dump's row order)
            ordered = {k: cols[k] for k in ("x", "y", "z")}
            ordered.update({k: v for k, v in cols.items() if k not in ordered})
            return Frame(ordered), Box(box, boundary, origin), info
        else:
            i += 1
    raise ValueError(f"{path}: no 'ITEM: ATOMS' section")


def read_file(path: str, fmt: Optional[str] = None):
    p = str(path)
    if fmt is None:
        base = p[:-3] if p.endswith(".gz") else p
        fmt = base.rsplit(".", 1)[-1].lower()
    if fmt == "xyz":
        return read_xyz(p)
    if fmt == "dump":
        return read_dump(p)
    raise ValueError(f"unsupported file format {fmt!r} (supported: xyz, dump)")
