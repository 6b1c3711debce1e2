"""Backend shim of the text tokenizer (csrc/text.hip): bytes of an atom table -> typed columns in HBM.

Two pieces: :func:`stream_to_device` moves a file's bytes into HBM through two pinned buffers (the next chunk is read from
the file while the previous one crosses PCIe), :func:`parse_table` runs ``mdh_parse_table`` on the resident text and
patches the few fields the device hands back (``float()`` / ``int()`` on the host — the reference's own conversion)."""
import ctypes

import numpy as np

from . import _lib
from .devarray import HArray, current_stream_ptr, torch

FLOAT, INT, STR, SKIP = 0, 1, 2, 3


class RedoOverflow(OverflowError):
    """more fields need the host converter than the hand-back buffer holds (a column of long strings, of nan): the caller
    reads the table with the host tokenizer instead"""

    def __init__(self, n):
        super().__init__(f"{n} fields need the host converter")
        self.n = n

_NP = {FLOAT: np.float64, INT: np.int32, STR: np.int64}
CHUNK = 32 << 20


def stream_to_device(f, nbytes=None, chunk=CHUNK):
    """read ``f`` (binary file object, positioned at the first byte wanted) to its end into one uint8 tensor in HBM.
    ``nbytes``: the size when known (one allocation); otherwise the pieces are concatenated at the end."""
    t = torch()
    pinned = [t.empty(chunk, dtype=t.uint8).pin_memory() for _ in range(2)]
    busy = [None, None]
    side = t.cuda.Stream()
    cur = t.cuda.current_stream()
    pieces, total = [], 0
    whole = t.empty(int(nbytes), dtype=t.uint8, device="cuda") if nbytes is not None else None
    side.wait_stream(cur)  # `whole` may be a block the current stream has only just released: the copies wait for that work
    k = 0
    while True:
        if busy[k] is not None:
            busy[k].synchronize()  # the copy out of this pinned buffer is done
        view = memoryview(pinned[k].numpy())
        got = f.readinto(view)
        if not got:
            break
        if whole is not None and total + got > whole.numel():  # the file grew: fall back to pieces
            pieces, whole = [whole[:total]], None
        with t.cuda.stream(side):
            if whole is not None:
                whole[total:total + got].copy_(pinned[k][:got], non_blocking=True)
            else:
                piece = pinned[k][:got].to("cuda", non_blocking=True)
                piece.record_stream(cur)  # allocated under the side stream, consumed (and freed) on the current one
                pieces.append(piece)
            ev = t.cuda.Event()
            ev.record(side)
        busy[k] = ev
        total += got
        k ^= 1
    side.synchronize()
    t.cuda.current_stream().wait_stream(side)
    if whole is not None:
        return whole[:total], total
    if not pieces:
        return t.empty(0, dtype=t.uint8, device="cuda"), 0
    return (pieces[0] if len(pieces) == 1 else t.cat(pieces)), total


def parse_table(text, nrows, kinds, redo_cap=1 << 16):
    """``text``: uint8 tensor in HBM holding the rows.  Returns (columns, lines_present): per kind an HArray (f64 / i32), a
    numpy object array of strings, or None for skipped columns."""
    t = torch()
    ncol = len(kinds)
    nbytes = int(text.numel())
    cols = [None if k == SKIP else HArray.empty((nrows,), _NP[k]) for k in kinds]
    ptrs = (ctypes.c_void_p * ncol)(*[None if c is None else c.data_ptr() for c in cols])
    kind_arr = np.asarray(kinds, dtype=np.int32)
    redo = HArray.empty((2 * redo_cap,), np.int64)
    status = np.zeros(4, dtype=np.int64)
    rc = _lib.lib().mdh_parse_table(int(text.data_ptr()), nbytes, _lib.DEVICE, int(nrows), ncol, kind_arr.ctypes.data, ptrs,
                                    redo.data_ptr(), int(redo_cap), status.ctypes.data, _lib.DEVICE, current_stream_ptr())
    _lib.check(rc)
    nredo, short, bad, lines = (int(v) for v in status)
    if lines < nrows and not short:  # (no kernel ran — an empty body — or the count fell short without a row noticing)
        short = nrows - lines
    if short:
        raise ValueError(f"expected {nrows} atom rows with {ncol} fields each; {short} rows are missing or too short")
    if nredo > redo_cap:
        raise RedoOverflow(nredo)
    long_strings = {}
    if nredo:
        # the fields the device handed back (strings over 8 bytes, nan / inf, integers spelled 3.0, 20-digit mantissas):
        # their bytes in ONE copy to the host, float() / int() there, the values back with one indexed write per column
        todo = redo.dev()[: 2 * nredo].cpu().numpy().reshape(-1, 2)
        where = todo[:, 0].astype(np.int64)
        off = (todo[:, 1] >> 16).astype(np.int64)
        ln = (todo[:, 1] & 0xFFFF).astype(np.int64)
        span = np.where(ln == 0xFFFF, 4096, ln)  # longer than the length field: up to the next blank, found on the host
        starts = np.concatenate([[0], np.cumsum(span)])[:-1]
        idx = (np.repeat(off - starts, span) + np.arange(int(span.sum()))).clip(max=max(nbytes - 1, 0))
        blob = text[t.from_numpy(idx).to(text.device)].cpu().numpy().tobytes()
        rows_of, vals_of = {}, {}
        for q in range(nredo):
            r, c = divmod(int(where[q]), ncol)
            raw = blob[int(starts[q]): int(starts[q] + span[q])]
            tok = (raw.split()[0] if ln[q] == 0xFFFF else raw).decode()
            if kinds[c] == STR:
                long_strings.setdefault(c, {})[r] = tok
            else:  # float() / int() raise ValueError for what they reject, as the reference's conversion does
                rows_of.setdefault(c, []).append(r)
                vals_of.setdefault(c, []).append(float(tok) if kinds[c] == FLOAT else int(tok))
        for c, rows in rows_of.items():
            dev = cols[c].dev()
            dev[t.as_tensor(rows, dtype=t.int64, device=dev.device)] = t.as_tensor(vals_of[c], dtype=dev.dtype, device=dev.device)
    out = [_unpack_strings(col, long_strings.get(c)) if k == STR else col for c, (k, col) in enumerate(zip(kinds, cols))]
    return out, lines


def _unpack_strings(packed, long_ones=None):
    """int64 of up to 8 little-endian characters -> numpy object array of str (unique codes decoded once)"""
    codes = packed.numpy()
    uniq, inv = np.unique(codes, return_inverse=True)
    names = np.array([int(u).to_bytes(8, "little", signed=True).rstrip(b"\0").decode() for u in uniq], dtype=object)
    out = names[inv]
    if long_ones:
        for r, tok in long_ones.items():
            out[r] = tok
    return out
