"""HBM-resident arrays and argument marshalling for the C ABI.

The reference hands numpy arrays to its extension modules.  On an MI355X the
neighbor arrays of a 10 M-atom system are ~2 GB; bouncing them over PCIe
between `build_neighbor` and every `cal_*` would dominate the run time, so the
host layer keeps them in HBM as :class:`HArray` (a torch ROCm tensor with a lazy,
read-only numpy mirror).  `HArray` quacks like the ndarray the reference exposes
(`shape`, indexing, `min`/`max`, ``np.asarray``), so user code and the parity
tests read `system.verlet_list[i, :n]` unchanged.

torch is used for device memory and streams only (plumbing); every kernel is in
mdapy_amd/csrc.
"""
from __future__ import annotations

import numpy as np

from . import _lib

_torch = None
_gpu = None

_NP2T = {}


def torch():
    global _torch
    if _torch is None:
        import torch as t  # deferred: `import torch` costs seconds and is not needed for host-only use

        _torch = t
        _NP2T.update({np.dtype(np.float64): t.float64, np.dtype(np.int32): t.int32, np.dtype(np.int64): t.int64,
                      np.dtype(np.float32): t.float32, np.dtype(np.uint8): t.uint8, np.dtype(np.int8): t.int8})
    return _torch


def have_gpu() -> bool:
    """True when a HIP device is visible to both the library and torch."""
    global _gpu
    if _gpu is None:
        try:
            _gpu = _lib.device_count() > 0 and torch().cuda.is_available()
        except Exception:
            _gpu = False
    return _gpu


_raw_stream = None


def current_stream_ptr() -> int:
    """the hipStream_t of torch's current stream on the current device (torch.cuda.current_stream() builds a Stream object and
    looks the device up three times: 7 us per call, twice per step of a 4 000-atom system whose whole step is 70)"""
    global _raw_stream
    t = torch()
    if _raw_stream is None:
        fast = getattr(t._C, "_cuda_getCurrentRawStream", None)
        _raw_stream = (lambda: int(fast(t.cuda.current_device()))) if fast is not None else (lambda: int(t.cuda.current_stream().cuda_stream))
    return _raw_stream()


PINNED_BUDGET = 1 << 30  # page-locked bytes HArray.numpy() may have handed out at any time (host arrays that are still alive)
_pinned_out = [0]


def _pinned_release(nbytes):
    _pinned_out[0] -= nbytes


class HArray:
    """An array that lives in HBM; the host copy is made on demand and is read-only."""

    __slots__ = ("_dev", "_host", "_mm", "__weakref__")
    __array_priority__ = 100

    def __init__(self, dev):
        self._dev = dev
        self._host = None
        self._mm = None  # (min, max) of an int32 array once somebody asked: dropped with the host copy (invalidate_host)

    # ---- construction
    @staticmethod
    def empty(shape, dtype):
        t = torch()
        return HArray(t.empty(tuple(int(s) for s in np.atleast_1d(shape)), dtype=_NP2T[np.dtype(dtype)], device="cuda"))

    @staticmethod
    def full(shape, value, dtype):
        t = torch()
        return HArray(t.full(tuple(int(s) for s in np.atleast_1d(shape)), value, dtype=_NP2T[np.dtype(dtype)], device="cuda"))

    @staticmethod
    def from_numpy(a):
        t = torch()
        a = np.ascontiguousarray(a)
        import warnings

        with warnings.catch_warnings():  # frame columns are read-only numpy views; the tensor is only read
            warnings.simplefilter("ignore")
            src = t.from_numpy(a)
        return HArray(src.to("cuda"))

    # ---- device side
    def dev(self):
        return self._dev

    def data_ptr(self) -> int:
        return int(self._dev.data_ptr())

    def invalidate_host(self):
        """call after a kernel changed the HBM content"""
        self._host = None
        self._mm = None

    # ---- host side
    def numpy(self) -> np.ndarray:
        if self._host is None:
            d = self._dev
            nbytes = d.numel() * d.element_size()
            locked = 1 << max(nbytes - 1, 1).bit_length()  # (torch's pinned allocator rounds a block up to a power of two)
            if d.is_cuda and d.is_contiguous() and (1 << 20) <= nbytes <= (1 << 28) and _pinned_out[0] + locked <= PINNED_BUDGET:
                # into page-locked memory (recycled by torch's host allocator) and handed out as it is: 0.7 instead of 5 ms for the
                # 40 MB label column of a 10 M-atom system.  The block stays locked while the host array lives, so the bytes
                # handed out this way are counted and capped (PINNED_BUDGET); beyond the cap a result is an ordinary pageable copy
                import weakref

                t = torch()
                pinned = t.empty(d.shape, dtype=d.dtype, pin_memory=True)
                pinned.copy_(d, non_blocking=True)
                t.cuda.current_stream().synchronize()
                h = pinned.numpy()
                _pinned_out[0] += locked
                weakref.finalize(pinned, _pinned_release, locked)
            else:
                h = d.cpu().numpy()
            h.setflags(write=False)
            self._host = mark_frozen(h)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    @property
    def shape(self):
        return tuple(self._dev.shape)

    @property
    def dtype(self):
        return np.dtype(str(self._dev.dtype).replace("torch.", ""))

    @property
    def ndim(self):
        return self._dev.dim()

    @property
    def size(self):
        return int(self._dev.numel())

    def __len__(self):
        return self._dev.shape[0]

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __iter__(self):
        return iter(self.numpy())

    def __repr__(self):
        return f"HArray(shape={self.shape}, dtype={self.dtype}, device='hbm')"

    def _min_max_i32(self):
        """(min, max) of an int32 array through the library's own reduction (mdh_min_max_i32): a torch reduction would do, but
        its first call in a process loads torch's reduction code object (19 ms on an MI355X box, tools/cold_profile.py)"""
        import ctypes

        if self._mm is None:  # (asked again and again for the counts of one list: every analysis checks the depth of the rows it borrows)
            d = self._dev if self._dev.is_contiguous() else self._dev.contiguous()
            out = (ctypes.c_int * 2)()
            _lib.check(_lib.lib().mdh_min_max_i32(int(d.data_ptr()), int(d.numel()), ctypes.addressof(out), _lib.DEVICE, current_stream_ptr()))
            self._mm = (int(out[0]), int(out[1]))
        return self._mm

    def min(self, *a, **k):
        if not a and not k and self.size:
            if self.dtype == np.int32:
                return self.dtype.type(self._min_max_i32()[0])
            return self.dtype.type(self._dev.min().item())
        return self.numpy().min(*a, **k)

    def max(self, *a, **k):
        if self.size and not a and set(k) <= {"initial"}:
            m = self._min_max_i32()[1] if self.dtype == np.int32 else self._dev.max().item()
            if "initial" in k:
                m = max(m, k["initial"])
            return self.dtype.type(m)
        return self.numpy().max(*a, **k)

    def sum(self, *a, **k):
        return self.numpy().sum(*a, **k)

    def copy(self):
        return self.numpy().copy()

    # ---- device-side reshaping for result columns (no host traffic)
    def column(self, j, dtype=None):
        """column j of a 2-D array as a contiguous 1-D array in HBM, optionally converted"""
        piece = self._dev[:, j]
        if dtype is not None:
            piece = piece.to(_NP2T[np.dtype(dtype)])
        return HArray(piece.contiguous())

    def head(self, n):
        """the first n rows (a view for 1-D and C-contiguous arrays)"""
        return self if n >= self._dev.shape[0] else HArray(self._dev[:n].contiguous())

    def astype(self, dtype, **k):
        return self.numpy().astype(dtype, **k)

    def tolist(self):
        return self.numpy().tolist()

    def _cmp(self, other, op):
        return op(self.numpy(), np.asarray(other))

    def __eq__(self, other):
        return self._cmp(other, np.equal)

    def __ne__(self, other):
        return self._cmp(other, np.not_equal)

    def __lt__(self, other):
        return self._cmp(other, np.less)

    def __le__(self, other):
        return self._cmp(other, np.less_equal)

    def __gt__(self, other):
        return self._cmp(other, np.greater)

    def __ge__(self, other):
        return self._cmp(other, np.greater_equal)

    __hash__ = None


class LazyHArray(HArray):
    """An HArray whose HBM content is produced when somebody first touches it (``make()`` -> torch tensor): the rows of a list
    that was built on the cell-sorted twin of a system, translated to the caller's atom order only if the caller reads them
    (system.py).  Shape and dtype are known beforehand and answer without producing anything."""

    __slots__ = ("_make", "_real", "_shape", "_np_dtype")

    def __init__(self, make, shape, dtype):
        self._make, self._real, self._host, self._mm = make, None, None, None
        self._shape, self._np_dtype = tuple(int(s) for s in shape), np.dtype(dtype)

    @property
    def _dev(self):
        if self._real is None:
            self._real = self._make()
            self._make = None
        return self._real

    @_dev.setter
    def _dev(self, value):
        self._real, self._make = value, None

    produced = property(lambda self: self._real is not None)
    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._np_dtype)
    ndim = property(lambda self: len(self._shape))
    size = property(lambda self: int(np.prod(self._shape)) if self._shape else 1)

    def __len__(self):
        return self._shape[0]


def as_numpy(a):
    """host ndarray view of numpy / HArray / Column input"""
    if isinstance(a, np.ndarray):
        return a
    if isinstance(a, HArray):
        return a.numpy()
    if hasattr(a, "to_numpy"):
        return a.to_numpy()
    return np.asarray(a)


def zeros(shape, dtype):
    """output buffer: HBM when a GPU is there, numpy otherwise (host-logic tests with a patched backend)"""
    if have_gpu():
        return HArray.full(shape, 0, dtype)
    return np.zeros(shape, dtype)


def full(shape, value, dtype):
    if have_gpu():
        return HArray.full(shape, value, dtype)
    return np.full(shape, value, dtype)


def empty(shape, dtype):
    if have_gpu():
        return HArray.empty(shape, dtype)
    return np.empty(shape, dtype)


# ---------------------------------------------------------------------------
# marshalling
# ---------------------------------------------------------------------------
import threading
import weakref

_cache_lock = threading.Lock()  # the C side takes mutexes for multi-threaded callers; so do the caches in front of it
_frozen = {}  # id -> (weak reference, fingerprint): arrays THIS package copied and froze (frame columns, cached species codes)
_mirrors = []  # [(weak reference to such an array, its copy in HBM)]


def _fingerprint(arr):
    """64 samples spread over the array: enough to notice that a frozen array was thawed, rewritten and frozen again"""
    flat = arr.reshape(-1)
    step = max(1, flat.shape[0] // 61)
    return (arr.shape, arr.dtype.str, flat[::step][:64].tobytes())


def mark_frozen(arr):
    """register a read-only array that the package itself made (only those are trusted not to change: identity-keyed caches —
    HBM mirrors here, species codes in policy — serve nobody else's arrays)"""
    try:
        with _cache_lock:
            if len(_frozen) > 256:
                for k in [k for k, (r, _) in _frozen.items() if r() is None]:
                    del _frozen[k]
            _frozen[id(arr)] = (weakref.ref(arr), _fingerprint(arr))
    except TypeError:
        pass
    return arr


def frozen_by_us(arr):
    with _cache_lock:
        hit = _frozen.get(id(arr))
    return hit is not None and hit[0]() is arr and not arr.flags.writeable and hit[1] == _fingerprint(arr)


def _mirror_of(arr):
    """HBM copy of a host array.  Large read-only arrays that the package froze itself (frame columns, the species codes
    policy.label_codes caches) do not change under us, so their copy is kept while the array lives and the next call that is
    handed the same array does not cross PCIe again."""
    if arr.nbytes < (1 << 20) or not arr.flags.owndata or not frozen_by_us(arr):  # (a read-only VIEW may still change through its base)
        return HArray.from_numpy(arr)
    with _cache_lock:
        for k in range(len(_mirrors) - 1, -1, -1):
            ref, dev = _mirrors[k]
            if ref() is None:
                del _mirrors[k]
            elif ref() is arr:
                return dev
    dev = HArray.from_numpy(arr)
    try:
        with _cache_lock:
            _mirrors.append((weakref.ref(arr), dev))
            del _mirrors[:-16]
    except TypeError:
        pass
    return dev



class Call:
    """Collects the array arguments of one C-ABI call and decides the memory space.

    space = HOST  iff every array argument is a plain numpy array (the library
                  stages through HBM itself);
    space = DEVICE otherwise: numpy inputs are uploaded, numpy outputs are
                  written back after the call.
    """

    def __init__(self, *arrays):
        self._keep = []
        self._writeback = []
        dev = False
        for a in arrays:
            if a is None or isinstance(a, np.ndarray):
                continue
            dev = True
        self.space = _lib.DEVICE if dev else _lib.HOST
        self.stream = current_stream_ptr() if dev else None

    def _is_dev(self):
        return self.space == _lib.DEVICE

    def inp(self, a, dtype):
        """read-only array argument -> pointer"""
        if a is None:
            return None
        dtype = np.dtype(dtype)
        if isinstance(a, HArray) or hasattr(a, "device_array"):
            h = a if isinstance(a, HArray) else a.device_array()
            if h.dtype != dtype:
                raise TypeError(f"expected {dtype}, got {h.dtype}")
            if not h.dev().is_contiguous():
                h = HArray(h.dev().contiguous())
            self._keep.append(h)
            return h.data_ptr()
        if _torch is not None and isinstance(a, _torch.Tensor):
            if not a.is_cuda:
                a = a.cuda()
            a = a.contiguous()
            if a.dtype != _NP2T[dtype]:
                a = a.to(_NP2T[dtype])
            self._keep.append(a)
            return int(a.data_ptr())
        arr = np.ascontiguousarray(as_numpy(a), dtype=dtype)  # read-only params accept convertible input (src/type.h:9-15)
        if self._is_dev():
            h = _mirror_of(arr)
            self._keep.append(h)
            return h.data_ptr()
        self._keep.append(arr)
        return arr.ctypes.data

    def out(self, a, dtype, upload=True):
        """writable array argument (exact dtype, C-contiguous, no implicit copy: src/type.h:16-21) -> pointer"""
        dtype = np.dtype(dtype)
        if isinstance(a, HArray):
            if a.dtype != dtype or not a.dev().is_contiguous():
                raise TypeError(f"writable argument must be C-contiguous {dtype}")
            a.invalidate_host()
            self._keep.append(a)
            return a.data_ptr()
        if _torch is not None and isinstance(a, _torch.Tensor):
            if not a.is_cuda or not a.is_contiguous() or a.dtype != _NP2T[dtype]:
                raise TypeError(f"writable tensor must be a contiguous ROCm tensor of {dtype}")
            self._keep.append(a)
            return int(a.data_ptr())
        if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous or not a.flags.writeable:
            raise TypeError(f"writable argument must be a writable C-contiguous numpy array of {dtype}")
        if self._is_dev():
            h = HArray.from_numpy(a) if upload else HArray.empty(a.shape, dtype)
            self._keep.append(h)
            self._writeback.append((a, h))
            return h.data_ptr()
        self._keep.append(a)
        return a.ctypes.data

    def done(self, rc):
        _lib.check(rc)
        for host, h in self._writeback:
            host[...] = h.dev().cpu().numpy()
        self._keep.clear()
