"""A minimal, immutable column frame standing in for the polars DataFrame the
reference stores per-atom data in (src/mdapy/system.py:275-294).

polars is not installed in this image; the hot path only needs: named numeric /
string columns of equal length, `frame["x"].to_numpy()`, `frame.select([...]).to_numpy()`,
`frame.with_columns(name=array)`, `frame.columns`, `frame.shape`.  A polars
DataFrame passed to `System(data=...)` is converted on entry.

Each numeric :class:`Column` can carry an HBM mirror (`device_array()`), uploaded
once and shared by every frame derived with `with_columns` — columns are
immutable (`to_numpy()` returns a read-only view), so the mirror cannot go stale.
"""
from __future__ import annotations

from typing import Dict, Iterable

import numpy as np


class Column:
    __slots__ = ("name", "_host_arr", "_dev", "__weakref__")  # (weakly referable: knn.py keeps candidate rows per position column)

    def __init__(self, name: str, values):
        self.name = name
        if type(values).__name__ in ("HArray", "LazyHArray"):  # already in HBM (the file readers): the host copy is made on first use
            if values.ndim != 1:
                raise ValueError(f"column {name!r} must be one-dimensional")
            self._host_arr = None
            self._dev = values
            return
        a = np.asarray(values)
        if a.ndim != 1:
            raise ValueError(f"column {name!r} must be one-dimensional")
        if a.dtype.kind in "iuf":
            a = np.ascontiguousarray(a)
        if a.flags.writeable:  # never freeze (or alias) the caller's own buffer
            from .devarray import mark_frozen

            a = a.copy()
            a.setflags(write=False)
            mark_frozen(a)
        self._host_arr = a
        self._dev = None

    @property
    def _host(self) -> np.ndarray:
        if self._host_arr is None:
            self._host_arr = self._dev.numpy()  # read-only
        return self._host_arr

    def to_numpy(self, allow_copy: bool = True, writable: bool = False) -> np.ndarray:
        if writable:
            return self._host.copy()
        return self._host

    def device_array(self):
        """HBM mirror (HArray), uploaded on first use"""
        if self._dev is None:
            from .devarray import HArray

            self._dev = HArray.from_numpy(self._host)
            self._dev._host = self._host
        return self._dev

    @property
    def dtype(self):
        return self._dev.dtype if self._host_arr is None else self._host_arr.dtype

    def __len__(self):
        return len(self._dev) if self._host_arr is None else self._host_arr.shape[0]

    def __array__(self, dtype=None, copy=None):
        a = self._host
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __getitem__(self, idx):
        return self._host[idx]

    def min(self):
        return self._host.min()

    def max(self):
        return self._host.max()

    def unique(self):
        return np.unique(self._host)

    def __repr__(self):
        return f"Column({self.name!r}, {self._host!r})"


class PermutedColumn(Column):
    """column ``source`` read through a permutation: element p is source[perm[p]].  Nothing moves until somebody asks — then
    in HBM (mdh_permute) when the permutation lives there, on the host otherwise.  (The per-atom columns of a system's
    cell-sorted twin, system.py.)"""
    __slots__ = ("_source", "_perm")

    def __init__(self, source: Column, perm):
        self.name = source.name
        self._source, self._perm = source, perm
        self._host_arr = None
        self._dev = None

    @property
    def _host(self) -> np.ndarray:
        if self._host_arr is None:
            if self._dev is None and self._in_hbm():
                self.device_array()  # (10 M numbers: a gather in HBM and one copy down, not a host gather through 40 MB of indices)
            if self._dev is not None:
                self._host_arr = self._dev.numpy()
            else:
                a = self._source.to_numpy()[np.asarray(self._perm)]
                a.setflags(write=False)
                self._host_arr = a
        return self._host_arr

    def _in_hbm(self):
        kind = np.dtype(self._source.dtype)
        return type(self._perm).__name__ in ("HArray", "LazyHArray") and kind.kind in "iuf" and kind.itemsize in (4, 8)

    def device_array(self):
        if self._dev is None:
            if self._in_hbm():
                from . import kernels

                self._dev = kernels.order.permute(self._source.device_array(), self._perm)
            else:
                from .devarray import HArray

                self._dev = HArray.from_numpy(self._host)
        return self._dev

    @property
    def dtype(self):
        return self._source.dtype

    def __len__(self):
        return len(self._source)


class Frame:
    def __init__(self, columns: Dict[str, Iterable] | None = None):
        self._cols: Dict[str, Column] = {}
        n = None
        for name, v in (columns or {}).items():
            c = v if isinstance(v, Column) else Column(name, v)
            if c.name != name:
                c = Column(name, c._dev if c._host_arr is None else c.to_numpy())
            if n is None:
                n = len(c)
            elif len(c) != n:
                raise ValueError(f"column {name!r} has length {len(c)}, expected {n}")
            self._cols[name] = c

    # ---- polars-like surface
    @property
    def columns(self):
        return list(self._cols)

    @property
    def shape(self):
        n = len(next(iter(self._cols.values()))) if self._cols else 0
        return (n, len(self._cols))

    @property
    def height(self):
        return self.shape[0]

    def __len__(self):
        return self.shape[0]

    def __contains__(self, name):
        return name in self._cols

    def __getitem__(self, name):
        if isinstance(name, str):
            return self._cols[name]
        if isinstance(name, (list, tuple)):
            return self.select(list(name))
        raise TypeError("Frame supports column access by name only")

    def select(self, *names):
        flat = []
        for n in names:
            flat.extend(n if isinstance(n, (list, tuple)) else [n])
        return Frame({n: self._cols[n] for n in flat})

    def drop(self, *names):
        flat = set()
        for n in names:
            flat.update(n if isinstance(n, (list, tuple)) else [n])
        return Frame({k: c for k, c in self._cols.items() if k not in flat})

    def with_columns(self, **new):
        cols = dict(self._cols)
        n = self.shape[0] if self._cols else None
        for name, v in new.items():
            if type(v).__name__ in ("HArray", "LazyHArray"):
                cols[name] = Column(name, v)
                continue
            a = v.to_numpy() if isinstance(v, Column) else np.asarray(v)
            if a.ndim == 0:
                a = np.full(n, a[()])
            cols[name] = Column(name, a)
        return Frame(cols)

    def to_numpy(self) -> np.ndarray:
        return np.column_stack([c.to_numpy() for c in self._cols.values()]) if self._cols else np.zeros((0, 0))

    def to_dict(self):
        return {k: c.to_numpy() for k, c in self._cols.items()}

    def rechunk(self):
        return self

    def filter(self, mask):
        mask = np.asarray(mask)
        return Frame({k: c.to_numpy()[mask] for k, c in self._cols.items()})

    def take(self, idx):
        return self.filter(idx)

    def __repr__(self):
        n, m = self.shape
        head = ", ".join(f"{k}:{c.dtype}" for k, c in self._cols.items())
        return f"Frame(shape=({n}, {m}); {head})"

    # ---- construction helpers
    @staticmethod
    def from_any(data) -> "Frame":
        if isinstance(data, Frame):
            return data
        if isinstance(data, dict):
            return Frame(data)
        if hasattr(data, "to_dict") and hasattr(data, "columns"):  # polars / pandas
            try:
                d = data.to_dict(as_series=False)  # polars
            except TypeError:
                d = {k: np.asarray(v) for k, v in data.to_dict(orient="list").items()}  # pandas
            return Frame({k: np.asarray(v) for k, v in d.items()})
        raise TypeError(f"cannot build a Frame from {type(data).__name__}")


def concat(frames):
    frames = list(frames)
    cols = frames[0].columns
    return Frame({c: np.concatenate([f[c].to_numpy() for f in frames]) for c in cols})
