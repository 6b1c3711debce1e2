"""Multi-GPU: slab domain decomposition with a ghost halo exchanged over RCCL (xGMI).

The reference is a single-process library; this layer is added by the build.  One process
per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm).  Atoms are owned by the rank whose slab (along one box axis,
in wrapped fractional coordinates) contains them; before a neighbor build every rank receives, from its two ring
neighbours, the atoms lying within `halo` of the shared faces (ncclSend / ncclRecv pairs — no other data-path collective).

Exactness.  The local problem is solved with the GLOBAL box: ghost positions are NOT shifted, the kernels apply the same
minimum-image arithmetic and use the same global cell grid as a single-GPU run, and the global ids order the atoms inside
a cell (``build_neighbor(..., key=gid)``).  The rows of owned atoms (ids, order, counts, distances) and every label derived
from them are therefore bit-identical to the single-GPU result for the whole system; rows of ghost atoms are incomplete and
discarded.

Transport.  With a process group that carries device tensors (nccl) the messages go GPU to GPU.  With "gloo" — the CPU
tests, and the `-m gpu` test that runs several ranks on ONE GPU — device tensors are staged through host memory; the code
path above the wire is the same.
"""
from dataclasses import dataclass, field

import numpy as np

from . import kernels
from .box import Box


def _torch():
    import torch

    return torch


@dataclass
class LocalDomain:
    """owned + ghost atoms of one rank, ordered by global id"""
    x: "object"
    y: "object"
    z: "object"
    gid: "object"       # int64 global ids, ascending
    owned: "object"     # bool mask
    n_owned: int
    extra: tuple = field(default_factory=tuple)  # further per-atom columns that travelled with the halo (e.g. types), f64


class SlabDecomposition:
    def __init__(self, box: Box, rank: int, world: int, axis: int = 0, group=None):
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside a world of {world}")
        if world > 1 and box.boundary[axis] != 1:
            raise ValueError("the decomposed axis must be periodic (ring of slabs)")
        self.box, self.rank, self.world, self.axis, self.group = box, rank, world, axis, group
        self.left = (rank - 1) % world
        self.right = (rank + 1) % world
        self._cap = 0  # rows of the halo pack buffers (grown on demand, sized from the last exchange)
        self._msg_cap = {}   # (halo, columns) -> atoms a halo message holds, agreed by all ranks (fast exchange)
        self._msg_buf = {}   # (columns, cap, device) -> list of buffer sets (four messages + pinned heads); one set per exchange in flight
        self._busy = set()   # id() of the buffer sets an exchange under way (a prefetched one included) is using
        self._roomy = {}     # data_ptr -> (weak reference, rows of spare room) of tensors made by with_room()
        self._own_mask = None
        self._side = None    # HIP stream the next frame's halo travels on while this frame's kernels run (start_halo)
        self._pending = {}   # (data_ptr of x, halo, columns) -> a halo exchange under way
        # what every step asks of the box, once (get_thickness is three numpy cross products: 0.2 ms of a 0.4 ms step's host time)
        self._thick = float(box.get_thickness()[axis])
        self._o = np.ascontiguousarray(box.origin, dtype=np.float64)
        self._hi3 = np.ascontiguousarray(box.inverse_box[:, axis], dtype=np.float64)

    # -- wire -----------------------------------------------------------------
    def _host_staged(self):
        import torch.distributed as dist

        return dist.get_backend(self.group) == "gloo"

    def _ring(self, to_right, to_left, from_left, from_right):
        """one exchange around the ring: two sends, two receives, batched"""
        import torch.distributed as dist

        stage = to_right.is_cuda and self._host_staged()
        sr, sl = (to_right.cpu(), to_left.cpu()) if stage else (to_right, to_left)
        rl, rr = (from_left.cpu(), from_right.cpu()) if stage else (from_left, from_right)
        ops = [dist.P2POp(dist.isend, sr, self.right, self.group), dist.P2POp(dist.isend, sl, self.left, self.group),
               dist.P2POp(dist.irecv, rl, self.left, self.group), dist.P2POp(dist.irecv, rr, self.right, self.group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if stage:
            from_left.copy_(rl)
            from_right.copy_(rr)

    def all_reduce_(self, tensor):
        """in-place sum over the ranks"""
        import torch.distributed as dist

        if self.world == 1:
            return tensor
        if tensor.is_cuda and self._host_staged():
            host = tensor.cpu()
            dist.all_reduce(host, group=self.group)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, group=self.group)
        return tensor

    # -- geometry -----------------------------------------------------------
    def frac(self, x, y, z):
        """wrapped fractional coordinate along the slab axis in [0, 1)"""
        t = _torch()
        hi = t.as_tensor(self.box.inverse_box[:, self.axis].copy(), dtype=t.float64, device=x.device)
        o = self.box.origin
        f = (x - o[0]) * hi[0] + (y - o[1]) * hi[1] + (z - o[2]) * hi[2]
        f = f - t.floor(f)
        return t.where(f >= 1.0, f - 1.0, f)

    def _select_device(self, x, y, z, up_from: float, down_below: float, gid=None):
        """-> (up, down) index tensors; with gid also the (n_sel, 4) rows (x, y, z, id as f64) of both selections"""
        import ctypes

        from . import _lib

        t = _torch()
        x, y, z = x.contiguous(), y.contiguous(), z.contiguous()
        assert x.dtype == t.float64 and y.dtype == t.float64 and z.dtype == t.float64
        n = int(x.shape[0])
        o = np.ascontiguousarray(self.box.origin, dtype=np.float64)
        hi3 = np.ascontiguousarray(self.box.inverse_box[:, self.axis], dtype=np.float64)
        cnt = (ctypes.c_int64 * 2)(0, 0)
        g = None
        if gid is not None:
            g = gid.contiguous()
            assert g.dtype == t.int64
        # A halo is a thin layer: the buffers hold what the last exchange needed plus a quarter (the first call guesses an
        # eighth of the slab); when a count comes back larger, the call is repeated with room for it.
        cap = min(n, self._cap if self._cap > 0 else max(1024, n // 8))
        while True:
            up = t.empty(cap, dtype=t.int32, device=x.device)
            down = t.empty(cap, dtype=t.int32, device=x.device)
            pu = pd = None
            if g is not None:
                pu = t.empty((cap, 4), dtype=t.float64, device=x.device)
                pd = t.empty((cap, 4), dtype=t.float64, device=x.device)
            _lib.check(_lib.lib().mdh_slab_halo_select(x.data_ptr(), y.data_ptr(), z.data_ptr(), n, o.ctypes.data, hi3.ctypes.data,
                                                       float(up_from), float(down_below), up.data_ptr(), down.data_ptr(), cnt,
                                                       g.data_ptr() if g is not None else None, pu.data_ptr() if pu is not None else None,
                                                       pd.data_ptr() if pd is not None else None, cap,
                                                       _lib.DEVICE, int(t.cuda.current_stream().cuda_stream)))
            nu, nd = int(cnt[0]), int(cnt[1])
            need = max(nu, nd)
            self._cap = min(n, need + need // 4 + 64)
            if need <= cap:
                break
            cap = self._cap
        if gid is None:
            return up[:nu].to(t.int64), down[:nd].to(t.int64)
        return up[:nu], down[:nd], pu[:nu], pd[:nd]

    def owner_of(self, x, y, z):
        t = _torch()
        return t.clamp((self.frac(x, y, z) * self.world).to(t.int64), 0, self.world - 1)

    def halo_fraction(self, halo: float) -> float:
        thick = self._thick
        h = (halo * (1.0 + 1e-9) + 1e-9) / thick
        if self.world > 1 and h > 1.0 / self.world + 1e-12:
            raise ValueError(f"slab thickness {thick / self.world:.3f} is smaller than the halo {halo}: use fewer ranks")
        return h

    def _single_message(self, h: float) -> bool:
        """may the single-message exchange serve this halo?  Three ranks and more: always.  Two ranks: the one neighbour is
        both the left and the right one, and an atom within `halo` of BOTH faces of its slab would arrive twice (the general
        path removes such duplicates); a slab at least four halos thick has no such atom."""
        return self.world > 2 or (self.world == 2 and h <= 0.125)

    # -- the fast exchange: one message per direction, one host read -------------------------------------------------
    def with_room(self, tensor, fraction: float = 0.25):
        """a copy of a 1-D tensor of owned atoms with spare rows behind it: exchange_halo(sort=False) then appends the ghosts
        in place instead of concatenating (a copy of every owned column per step otherwise).  Only tensors made here are
        ever written behind their end."""
        import weakref

        t = _torch()
        n = int(tensor.shape[0])
        room = max(1024, int(n * fraction))
        buf = t.empty(n + room, dtype=tensor.dtype, device=tensor.device)
        buf[:n].copy_(tensor)
        out = buf[:n]
        self._roomy[out.data_ptr()] = (weakref.ref(out), room)
        return out

    def _room_behind(self, tensor):
        hit = self._roomy.get(tensor.data_ptr())
        return hit[1] if hit is not None and hit[0]() is tensor else 0

    def _owned_mask(self, n_tot, n_owned, dev):
        hit = self._own_mask
        if hit is None or hit[0] != n_tot or hit[1] != n_owned or hit[2] != dev:
            hit = self._own_mask = (n_tot, n_owned, dev, _torch().arange(n_tot, device=dev) < n_owned)
        return hit[3]

    def reset_halo_capacity(self):
        """forget the agreed message sizes (call on all ranks; the next exchange agrees on new ones)"""
        self._drop_pending()
        self._msg_cap.clear()
        self._msg_buf.clear()

    def _drop_pending(self):
        """forget exchanges that were started and never picked up; their buffer sets are free again once the streams that
        filled them are waited for (the next exchange on either stream is ordered behind them)"""
        t = _torch()
        for st in self._pending.values():
            if st["done"] is not None:
                st["done"].synchronize()
            if st["side"]:
                t.cuda.current_stream().wait_event(st["done"])
            self._busy.discard(id(st["bufs"]))
        self._pending.clear()

    def _exchange_fast(self, x, y, z, gid, cols, h, halo, static=False):
        """device tensors, ghosts appended behind the owned atoms.  The layers are selected and packed into the two outgoing
        messages on the device (slab.hip), each message carries its atom count in its first word, and the host reads the four
        counts — sent and received — in ONE copy after the ring: no count round trip, no second synchronisation.  The message
        size is agreed by all ranks (largest layer + 25 %) the first time a (halo, columns) pair is seen.

        static=True (and room for two whole messages behind every column): the host reads NOTHING — the ghost block has the
        fixed size 2 cap, the counts stay in the message headers on the device, unused slots are marked absent (x = NaN, id -1;
        slab.hip k_slab_append_static) and a message that did not fit is reported by the next exchange (or check_halo())."""
        key = (x.data_ptr(), float(halo), len(cols))
        state = self._pending.pop(key, None)
        if state is None or state["n_owned"] != int(x.shape[0]):
            state = self._fast_begin(x, y, z, gid, cols, h, halo, side=False, heads=not static)
        return self._fast_end(state, static)

    def check_halo(self, whose="a"):
        """raise if a halo message of an exchange with static=True that has RUN on the device did not fit its agreed size (its
        ghost block was cut short: border atoms of that step have wrong rows and labels).  A static step cannot know this
        about itself without waiting for the device, so the flag of step n is seen by step n + 1 — and the flag of the LAST
        step of a run only by this call: synchronise the stream and call it once after the last step (neighbor_cna_step(...,
        strict=True) does both)"""
        from . import _lib

        if _lib.lib().mdh_slab_overflow_check() != 0:
            raise RuntimeError(whose + " halo message did not fit the agreed size (" + _lib.lib().mdh_last_error().decode("utf-8", "replace") + "): the system changed since the "
                               "size was agreed — call reset_halo_capacity() on all ranks and repeat the step")

    def start_halo(self, x, y, z, gid, halo: float, extra=(), static=False):
        """Begin the halo exchange of a frame — selection, packing, the ring, the read of the counts — on a side stream, and
        return at once: the kernels of the frame before keep the device busy meanwhile (frames of a trajectory do not depend
        on each other).  The exchange_halo(x, y, z, gid, halo, sort=False, extra=...) that follows with the SAME tensors picks
        the messages up; until then the owned atoms must not change.  Without the single-message exchange (two ranks, host
        tensors) this does nothing and exchange_halo does the whole exchange; with a host-staged ring (gloo) the exchange is
        complete when this returns."""
        t = _torch()
        if self.world < 2 or not x.is_cuda or len(extra) > 4 or not self._single_message(self.halo_fraction(halo)):
            return
        cols = [x, y, z] + [e.to(t.float64) for e in extra]
        key = (x.data_ptr(), float(halo), len(cols))
        if key in self._pending:
            return
        if len(self._pending) > 2:
            self._drop_pending()
        self._pending[key] = self._fast_begin(x.contiguous(), y.contiguous(), z.contiguous(), gid, cols, self.halo_fraction(halo), halo, side=True,
                                              heads=not static)

    def _fast_begin(self, x, y, z, gid, cols, h, halo, side, heads=True):
        import ctypes

        from . import _lib

        t = _torch()
        dev = x.device
        n_owned = int(x.shape[0])
        width = len(cols) + 1
        lo, hi = self.rank / self.world, (self.rank + 1) / self.world
        sig = (float(halo), width)  # (the same on every rank: all ranks agree on a size in the same call)
        cap = self._msg_cap.get(sig)
        if cap is None:  # agree on a size: every rank's largest layer, one all-reduce, once per signature
            up, down = self._select_device(x, y, z, hi - h, lo + h)
            need = t.tensor([max(int(up.shape[0]), int(down.shape[0]))], dtype=t.int64, device=dev)
            import torch.distributed as dist

            if need.is_cuda and self._host_staged():
                need = need.cpu()
            dist.all_reduce(need, op=dist.ReduceOp.MAX, group=self.group)
            cap = int(need.item())
            cap = cap + cap // 4 + 64
            self._msg_cap[sig] = cap
        key = (width, cap, str(dev))
        if key not in self._msg_buf and len(self._msg_buf) > 4:
            for k in [k for k, sets in self._msg_buf.items() if not any(id(b) in self._busy for b in sets)]:
                del self._msg_buf[k]
        pool = self._msg_buf.setdefault(key, [])
        # an exchange owns its buffer set — the four messages and the pinned heads — from here until _fast_end has read it: a
        # prefetched exchange (start_halo) that is still waiting to be picked up keeps its set, and any other exchange with
        # the same (halo, columns) in between takes another one
        bufs = next((b for b in pool if id(b) not in self._busy), None)
        if bufs is None:
            bufs = tuple(t.empty(1 + width * cap, dtype=t.float64, device=dev) for _ in range(4)) + (t.empty(4, dtype=t.float64).pin_memory(),)
            pool.append(bufs)
        self._busy.add(id(bufs))
        send_r, send_l, recv_l, recv_r, heads_pinned = bufs
        o, hi3 = self._o, self._hi3
        ex = [c.contiguous() for c in cols[3:]]
        exp = (ctypes.c_void_p * max(len(ex), 1))(*[e.data_ptr() for e in ex]) if ex else None
        g = gid.contiguous()
        main = t.cuda.current_stream()
        if side:
            if self._side is None:
                self._side = t.cuda.Stream(device=dev)
            stream = self._side
            stream.wait_stream(main)  # the owned atoms are ready, and the previous frame's ghosts have been read out of the receive buffers
        else:
            stream = main
        try:
            import contextlib

            with (t.cuda.stream(stream) if side else contextlib.nullcontext()):  # (switching to the stream that is current costs ~20 us of host time)
                _lib.check(_lib.lib().mdh_slab_halo_messages(x.data_ptr(), y.data_ptr(), z.data_ptr(), n_owned, o.ctypes.data, hi3.ctypes.data,
                                                             float(hi - h), float(lo + h), g.data_ptr(), exp, len(ex), send_r.data_ptr(),
                                                             send_l.data_ptr(), cap, int(stream.cuda_stream)))
                self._ring(send_r, send_l, recv_l, recv_r)
                if heads:  # the one device-to-host read (static exchanges leave the counts on the device)
                    heads_pinned.copy_(t.stack([send_r[0], send_l[0], recv_l[0], recv_r[0]]), non_blocking=True)
                done = None
                if side or heads:  # (an exchange on the main stream that nobody waits for needs no event)
                    done = t.cuda.Event()
                    done.record(stream)
        except BaseException:
            self._busy.discard(id(bufs))  # (an exchange that never got under way does not keep its buffer set)
            raise
        return dict(cols=cols, gid=gid, ex=ex, g=g, n_owned=n_owned, cap=cap, sig=sig, bufs=bufs, done=done, side=side, dev=dev, heads=heads)

    def _fast_end(self, st, static=False):
        t = _torch()
        if static and not st["heads"] and self._fast_end_static_ok(st):
            return self._fast_end_static(st)
        if st["done"] is not None:
            st["done"].synchronize()  # (waits for the exchange only: kernels of the previous frame on the main stream keep running)
        if st["side"]:
            t.cuda.current_stream().wait_event(st["done"])
        send_r, send_l, recv_l, recv_r, heads_pinned = st["bufs"]
        if not st["heads"]:  # begun as a static exchange, ended as a plain one: read the counts now
            heads_pinned.copy_(t.stack([send_r[0], send_l[0], recv_l[0], recv_r[0]]))
        heads = heads_pinned.tolist()
        self._busy.discard(id(st["bufs"]))  # (what reads the receive buffers below is enqueued before any later exchange refills them)
        cap, dev, n_owned, cols, gid = st["cap"], st["dev"], st["n_owned"], st["cols"], st["gid"]
        if max(heads) > cap:
            # only the two ranks of the overflowing link see this: the agreed size is NOT dropped here (a subset of ranks
            # re-agreeing would enter an all-reduce the others never join); the job stops, or every rank resets together
            raise RuntimeError(f"halo message of {int(max(heads))} atoms does not fit the agreed {cap}: the system changed since the "
                               "size was agreed — call reset_halo_capacity() on all ranks and repeat the step")
        nl, nr = int(heads[2]), int(heads[3])
        n_ghost = nl + nr
        n_tot = n_owned + n_ghost

        def row(msg, r, cnt):
            return msg[1 + r * cap: 1 + r * cap + cnt]

        every = list(cols) + [gid]
        if all(self._room_behind(c) >= n_ghost for c in every) and all(c.dtype == t.float64 for c in cols) and gid.dtype == t.int64 and len(cols) <= 7:
            # every column has room behind its owned atoms: one kernel writes all ghosts in place
            import ctypes

            from . import _lib

            ptrs = (ctypes.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
            _lib.check(_lib.lib().mdh_slab_append_ghosts(recv_l.data_ptr(), recv_r.data_ptr(), cap, nl, nr, ptrs, len(cols), gid.data_ptr(), n_owned,
                                                         int(t.cuda.current_stream().cuda_stream)))
            full = [t.empty(0, dtype=c.dtype, device=dev).set_(c.untyped_storage(), c.storage_offset(), (n_tot,), (1,)) for c in every]
            return LocalDomain(full[0], full[1], full[2], full[-1], self._owned_mask(n_tot, n_owned, dev), n_owned, tuple(full[3:-1]))
        full = []
        for k, c in enumerate(every):
            if self._room_behind(c) >= n_ghost:  # ghosts written behind the owned atoms, in place
                whole = t.empty(0, dtype=c.dtype, device=dev).set_(c.untyped_storage(), c.storage_offset(), (n_tot,), (1,))
            else:
                whole = t.empty(n_tot, dtype=c.dtype, device=dev)
                whole[:n_owned].copy_(c)
            whole[n_owned:n_owned + nl].copy_(row(recv_l, k, nl))  # (the id row converts f64 -> i64 in the copy)
            whole[n_owned + nl:].copy_(row(recv_r, k, nr))
            full.append(whole)
        return LocalDomain(full[0], full[1], full[2], full[-1], self._owned_mask(n_tot, n_owned, dev), n_owned, tuple(full[3:-1]))

    def _fast_end_static_ok(self, st):
        t = _torch()
        cols, gid, cap = st["cols"], st["gid"], st["cap"]
        return (all(self._room_behind(c) >= 2 * cap for c in list(cols) + [gid]) and all(c.dtype == t.float64 for c in cols)
                and gid.dtype == t.int64 and len(cols) <= 7)

    def _fast_end_static(self, st):
        """no host read, no host wait: the ghost block is 2 cap slots, filled from the headers on the device"""
        import ctypes

        from . import _lib

        t = _torch()
        self._busy.discard(id(st["bufs"]))  # (before anything can raise: a failed check must not keep the buffer set marked busy)
        self.check_halo("an earlier step's")  # (an overflow of an EARLIER step that has run by now)
        if st["side"]:
            t.cuda.current_stream().wait_event(st["done"])
        send_r, send_l, recv_l, recv_r, _ = st["bufs"]
        cap, dev, n_owned, cols, gid = st["cap"], st["dev"], st["n_owned"], st["cols"], st["gid"]
        ptrs = (ctypes.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
        _lib.check(_lib.lib().mdh_slab_append_ghosts_static(recv_l.data_ptr(), recv_r.data_ptr(), cap, ptrs, len(cols), gid.data_ptr(), n_owned,
                                                            int(t.cuda.current_stream().cuda_stream)))
        n_tot = n_owned + 2 * cap
        full = [t.empty(0, dtype=c.dtype, device=dev).set_(c.untyped_storage(), c.storage_offset(), (n_tot,), (1,)) for c in list(cols) + [gid]]
        dom = LocalDomain(full[0], full[1], full[2], full[-1], self._owned_mask(n_tot, n_owned, dev), n_owned, tuple(full[3:-1]))
        dom.absent_slots = True  # rows [n_owned, n_tot) hold ghosts and absent atoms (x = NaN, gid = -1)
        return dom

    def hint_window(self, halo: float, like=None, own_rows_only: bool = False):
        """tell the next neighbor build where this rank's atoms are (slab + halo along the decomposed axis): its passes over the
        cells of the GLOBAL grid then cover that window only.  own_rows_only: the build makes no rows for the ghosts (they are listed
        as neighbours of the owned atoms, which is what a step needs them for) — counts of ghost atoms stay as the caller zeroed them"""
        if self.world >= 2 and (like is None or getattr(like, "is_cuda", False)) and hasattr(kernels.neighbor, "hint_cell_window"):
            h = self.halo_fraction(halo)
            if not self._single_message(h):
                return
            kernels.neighbor.hint_cell_window(self.axis, self.rank / self.world - h, (self.rank + 1) / self.world + h)
            if own_rows_only and hasattr(kernels.neighbor, "hint_centre_window"):
                kernels.neighbor.hint_centre_window(self.axis, self.rank / self.world, (self.rank + 1) / self.world)

    # -- halo exchange --------------------------------------------------------
    def exchange_halo(self, x, y, z, gid, halo: float, sort: bool = True, extra=(), static: bool = False) -> LocalDomain:
        """x,y,z (f64) and gid (i64) of the OWNED atoms (1-D tensors on this rank's device).

        sort=True: the local order is ascending global id (what index-ordered kernels need to reproduce the undivided
        system's rows).  sort=False: owned atoms first, in the caller's order, then the ghosts — for kernels that take the
        ids as an ordering key (``build_neighbor(..., key=dom.gid)``); no pass over the owned atoms beyond the copy.
        extra: further per-owned-atom columns (numbers exact in f64, e.g. int32 types); they travel in the same message
        as the positions and come back as ``dom.extra`` in local order.
        static (with sort=False, tensors from with_room() with room for two whole messages): nothing is read by the host; the
        domain then has a ghost block of fixed size whose unused slots are ABSENT atoms (x = NaN, gid = -1, ``dom.absent_slots``)
        — only kernels that skip them may take it (the neighbor build does: mdh_build_neighbor*, include/mdapy_amd.h)."""
        t = _torch()
        n_owned = int(x.shape[0])
        dev = x.device
        cols = [x, y, z] + [e.to(t.float64) for e in extra]  # everything that follows an atom (besides its id)
        if self.world == 1:
            if n_owned and not bool((gid[1:] > gid[:-1]).all()):
                order = t.argsort(gid)
                cols, gid = [c[order] for c in cols], gid[order]
            return LocalDomain(cols[0], cols[1], cols[2], gid, t.ones(n_owned, dtype=t.bool, device=dev), n_owned, tuple(cols[3:]))
        h = self.halo_fraction(halo)
        lo, hi = self.rank / self.world, (self.rank + 1) / self.world
        if x.is_cuda and not sort and len(extra) <= 4 and self._single_message(h):
            return self._exchange_fast(x.contiguous(), y.contiguous(), z.contiguous(), gid, cols, h, halo, static)
        if x.is_cuda:  # selection and packing in one fused pass (slab.hip); the torch expressions below are its definition
            up, down, rows_r, rows_l = self._select_device(x, y, z, hi - h, lo + h, gid)
            send_r, send_l = rows_r.t(), rows_l.t()  # rows x, y, z, id
            if extra:
                iu, idn = up.long(), down.long()
                send_r = t.cat([send_r[:3]] + [c[iu][None] for c in cols[3:]] + [send_r[3:]])
                send_l = t.cat([send_l[:3]] + [c[idn][None] for c in cols[3:]] + [send_l[3:]])
            send_r, send_l = send_r.contiguous(), send_l.contiguous()
        else:
            f = self.frac(x, y, z)
            up = (f >= hi - h).nonzero().flatten()    # goes to the right neighbour
            down = (f < lo + h).nonzero().flatten()   # goes to the left neighbour

            def pack(sel):
                return t.stack([c[sel] for c in cols] + [gid[sel].to(t.float64)], dim=0).contiguous()  # ids < 2^53: exact

            send_r, send_l = pack(up), pack(down)
        width = len(cols) + 1
        # sizes first, then the payload
        cnt_in = t.zeros(2, dtype=t.int64, device=dev)
        self._ring(t.tensor([send_r.shape[1]], dtype=t.int64, device=dev), t.tensor([send_l.shape[1]], dtype=t.int64, device=dev),
                   cnt_in[0:1], cnt_in[1:2])
        n_from_left, n_from_right = (int(v) for v in cnt_in.tolist())  # one device-to-host read for both counts
        recv_l = t.empty((width, n_from_left), dtype=t.float64, device=dev)
        recv_r = t.empty((width, n_from_right), dtype=t.float64, device=dev)
        self._ring(send_r, send_l, recv_l, recv_r)
        ghosts = t.cat([recv_l, recv_r], dim=1)
        ggid = ghosts[-1].to(t.int64)
        if sort or self.world == 2:
            # ghosts in ascending id order, each once (world == 2: the same atom can arrive through both faces of the one neighbour)
            ggid, first = _unique_first(ggid)
            ghosts = ghosts[:, first]
        n_ghost = int(ggid.shape[0])
        n_tot = n_owned + n_ghost
        if not sort:  # owned first, then the ghosts
            own = t.arange(n_tot, device=dev) < n_owned
            full = [t.cat([c, ghosts[k]]) for k, c in enumerate(cols)]
            return LocalDomain(full[0], full[1], full[2], t.cat([gid, ggid]), own, n_owned, tuple(full[3:]))
        if n_owned < 2 or bool((gid[1:] > gid[:-1]).all()):
            # Local order = ascending global id.  The owned ids are already ascending, the few ghosts are sorted: MERGE the two
            # runs (two binary searches, two scatters per array) instead of sorting all of them every step.
            gslot = t.searchsorted(gid, ggid) + t.arange(n_ghost, dtype=t.int64, device=dev)
            oslot = t.searchsorted(ggid, gid) + t.arange(n_owned, dtype=t.int64, device=dev)

            def merge(a_owned, a_ghost):
                out = t.empty(n_tot, dtype=a_owned.dtype, device=dev)
                out[oslot] = a_owned
                out[gslot] = a_ghost.to(a_owned.dtype)
                return out

            own = t.zeros(n_tot, dtype=t.bool, device=dev)
            own[oslot] = True
            full = [merge(c, ghosts[k]) for k, c in enumerate(cols)]
            return LocalDomain(full[0], full[1], full[2], merge(gid, ggid), own, n_owned, tuple(full[3:]))
        ag = t.cat([gid, ggid])
        order = t.argsort(ag)
        own = (t.arange(n_tot, device=dev) < n_owned)[order].contiguous()
        full = [t.cat([c, ghosts[k]])[order].contiguous() for k, c in enumerate(cols)]
        return LocalDomain(full[0], full[1], full[2], ag[order].contiguous(), own, n_owned, tuple(full[3:]))


def _unique_first(ids):
    t = _torch()
    s, idx = t.sort(ids, stable=True)
    keep = t.ones_like(s, dtype=t.bool)
    keep[1:] = s[1:] != s[:-1]
    return s[keep], idx[keep]


def partition_atoms(pos: np.ndarray, box: Box, world: int, axis: int = 0):
    """host helper: global ids owned by every rank (list of int64 arrays) for arbitrary input order"""
    f = (pos - box.origin) @ box.inverse_box[:, axis]
    f = f - np.floor(f)
    f[f >= 1.0] -= 1.0
    owner = np.clip((f * world).astype(np.int64), 0, world - 1)
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world)]


def neighbor_cna_step(dec: SlabDecomposition, x, y, z, gid, rc: float, max_neigh: int, next_frame=None, strict: bool = False,
                      reuse_buffers: bool = False):
    """One pass of the distributed hot path: halo exchange -> neighbor build -> fixed-cutoff CNA.

    Returns (dom, verlet, dist, nn, pattern): neighbor arrays / labels in `dom` order — the rows of the OWNED atoms
    (``dom.owned``).  What the arrays hold for a GHOST is unspecified: ghosts are listed as neighbours, nobody needs THEIR
    neighbours, and the tile kernel makes no rows for the cell planes that hold ghosts only (pads, count 0, label 0 there; a ghost
    in a plane it shares with owned atoms gets its — by nature incomplete — row).  ``dec._ghost_rows = True`` makes all ghost rows as
    earlier rounds did; on the CPU path they are always made.  ``verlet`` holds local indices; ``dom.gid[verlet]`` maps them to
    global ids.

    next_frame = (x, y, z, gid) of the frame the NEXT call will be given: its halo exchange is started on a side stream
    before this frame's kernels are enqueued and travels while they run (SlabDecomposition.start_halo).
    strict: wait for the build and raise if an atom lay outside the slab + halo window the build was promised (atoms that
    drifted across a face without re-partitioning) or if THIS step's halo message did not fit its agreed size; without it
    the build is still memory-safe, the NEXT step raises for this one, and the caller must end a run with
    ``torch.cuda.synchronize(); dec.check_halo()`` (the last step has no successor to report it).

    reuse_buffers: the four result arrays are the SAME tensors from call to call (kept on `dec`, as a caller who allocates his
    outputs once would have them — bench.py's single-GPU step does): a step then starts with one fill (the labels) instead of four —
    the ghosts' pads and zero counts of the previous step are still there, no kernel writes those rows.  The previous step's results
    are overwritten.

    On more than one rank with HBM-resident tensors the exchange is *static*: the ghost count never leaves the device, so
    the local arrays have ``n_owned + 2 * cap`` rows (``dom.absent_slots`` is True) of which the slots behind the ghosts
    that arrived hold ABSENT atoms: ``x = NaN``, ``dom.gid = -1``, ``nn = 0``, a row of pads (-1 / rc + 1), label 0.  Index
    with ``dom.owned`` (or ``dom.gid >= 0``) before scattering by ``dom.gid``.  ``dec._no_static = True`` switches back to
    the exchange that reads the counts (one host wait per step, exact-size arrays).
    """
    t = _torch()
    # static: the ghost count stays on the device (a ghost block of fixed size with absent slots) — nothing in this step waits
    # for the device, the host runs ahead of it
    static = bool(getattr(x, "is_cuda", False)) and dec.world > 1 and not getattr(dec, "_no_static", False)
    dom = dec.exchange_halo(x, y, z, gid, rc, sort=False, static=static)
    if next_frame is not None and not getattr(dec, "_no_prefetch", False):
        try:
            dec.start_halo(*next_frame, rc, static=static)
        except Exception as e:  # the overlap is an optimisation: without it the next call exchanges inside the step
            import sys

            dec._no_prefetch = True
            dec._drop_pending()
            print(f"mdapy_amd.distributed: halo prefetch switched off on rank {dec.rank} ({type(e).__name__}: {e})", file=sys.stderr)
    n = int(dom.x.shape[0])
    b = dec.box
    # (absent slots get no row: their counts must read 0 for the CNA behind the build)
    ghost_rows = bool(getattr(dec, "_ghost_rows", False))  # (True: rows for the ghosts as well, as rounds 3-5 made them; incomplete by nature)
    skip_ghosts = dec.world > 1 and not ghost_rows and bool(getattr(dom.x, "is_cuda", False))
    kept = getattr(dec, "_step_buffers", None) if reuse_buffers else None
    sig = (n, int(max_neigh), dom.n_owned, float(rc), str(dom.x.device), skip_ghosts)
    if kept is not None and kept[0] == sig and skip_ghosts:
        verlet, dist, nn, pattern = kept[1]
        pattern.zero_()  # (the kernels only ever raise a label; the ghosts' pads and zero counts are the previous step's, untouched)
    else:
        verlet = t.empty((n, max_neigh), dtype=t.int32, device=dom.x.device)
        dist = t.empty((n, max_neigh), dtype=t.float64, device=dom.x.device)
        nn = (t.zeros if (getattr(dom, "absent_slots", False) or skip_ghosts) else t.empty)((n,), dtype=t.int32, device=dom.x.device)
        pattern = t.zeros((n,), dtype=t.int32, device=dom.x.device)
        if getattr(dom, "absent_slots", False) or skip_ghosts:  # the build writes no row for an absent atom or a ghost: pads, not what the allocator left
            verlet[dom.n_owned:].fill_(-1)
            dist[dom.n_owned:].fill_(rc + 1.0)
        if reuse_buffers:
            dec._step_buffers = (sig, (verlet, dist, nn, pattern))
    dec.hint_window(rc, dom.x, own_rows_only=skip_ghosts)
    # lists and labels in ONE pass over the tiles (mdh_build_neighbor_fcna: a centre's 12 or 14 neighbours are still staged in LDS
    # when its row is written) — bit for bit what build_neighbor followed by fcna leaves
    kernels.neighbor.build_neighbor_fcna(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, rc, verlet, dist, nn, pattern, 1, fill_pads=True,
                                         key=dom.gid if dec.world > 1 else None)
    if strict and dom.x.is_cuda and hasattr(kernels.neighbor, "cell_window_check"):
        kernels.neighbor.cell_window_check()
        if static:
            t.cuda.current_stream().synchronize()
            dec.check_halo("this step's")
    return dom, verlet, dist, nn, pattern


# ------------------------------------------------------------------------------------------------------------------
# k-nearest-neighbour analyses (adaptive CNA, CSP, PTM, ...): the halo is not known in advance
# ------------------------------------------------------------------------------------------------------------------
def knn_step(dec: SlabDecomposition, x, y, z, gid, k: int, halo=None, neighbor_rows: int = 0, max_tries: int = 6, extra=()):
    """k nearest neighbours of every owned atom, bit-identical to a single-GPU search of the whole system.

    A slab knows the atoms within `halo` of its faces.  The k-th neighbour distance r_k(i) of an atom proves its own row
    complete iff r_k(i) is smaller than the distance from i to the edge of the known region; the halo is doubled (all
    ranks together, one 1-element all-reduce per try) until that holds for every owned atom — and, when
    ``neighbor_rows = m > 0``, also for the first m neighbours of every owned atom (analyses that read their
    neighbours' rows: identify-diamond m=4, the two-shell PTM templates m=13).

    Returns (dom, idx, dist, valid): rows in ``dom`` order; ``idx`` holds local indices (``dom.gid[idx]`` = global ids);
    ``valid`` marks the rows proven complete (all owned rows are).
    """
    t = _torch()
    b = dec.box
    thick = float(b.get_thickness()[dec.axis])
    if halo is None:  # about the k-th neighbour distance of a uniform system of the local density, with head-room
        n_loc = max(int(x.shape[0]), 1)
        vol = float(b.volume) / dec.world
        halo = 1.5 * (3.0 * (k + 1) / (4.0 * np.pi) * vol / n_loc) ** (1.0 / 3.0)
    for _ in range(max_tries):
        if dec.world > 1:
            halo = min(halo, thick / dec.world * (1.0 - 1e-6))
        dom = dec.exchange_halo(x, y, z, gid, halo, extra=extra)
        n = int(dom.x.shape[0])
        idx = t.empty((n, k), dtype=t.int32, device=dom.x.device)
        dst = t.empty((n, k), dtype=t.float64, device=dom.x.device)
        kernels.fast_knn.knn(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, k, idx, dst, 1)
        if dec.world == 1:
            return dom, idx, dst, t.ones(n, dtype=t.bool, device=dom.x.device)
        h = dec.halo_fraction(halo)
        f = dec.frac(dom.x, dom.y, dom.z)
        lo, hi = dec.rank / dec.world, (dec.rank + 1) / dec.world
        # fractional distance to the nearer edge of the known region [lo - h, hi + h), measured around the ring
        centre = 0.5 * (lo + hi)
        off = f - centre
        off = off - t.round(off)
        room = (0.5 * (hi - lo) + h - off.abs()) * thick
        if (hi - lo) + 2 * h >= 1.0 - 1e-12:
            room = t.full_like(room, float("inf"))  # own slab + both halos cover the whole ring
        valid = dst[:, k - 1] < room * (1.0 - 1e-12)
        ok = valid[dom.owned]
        if neighbor_rows > 0:
            m = min(neighbor_rows, k)
            nb = idx[dom.owned][:, :m].long()
            ok = ok & valid[nb.clamp(min=0)].all(dim=1)
        bad = dec.all_reduce_(t.tensor([int((~ok).sum().item())], dtype=t.int64, device=dom.x.device))
        if int(bad.item()) == 0:
            return dom, idx, dst, valid
        if halo >= thick / dec.world * (1.0 - 1e-6):
            raise RuntimeError(f"knn_step: slab of thickness {thick / dec.world:.3f} cannot prove {k} neighbours complete; "
                               "use fewer ranks along this axis")
        halo *= 2.0
    raise RuntimeError("knn_step: halo did not converge")


def knn_analysis_step(dec: SlabDecomposition, x, y, z, gid, what=("acna", "csp", "ptm"), csp_neighbors: int = 12,
                      ptm_structure: str = "fcc-hcp-bcc", ptm_threshold: float = 0.1, types=None):
    """Adaptive CNA / CSP / PTM of the owned atoms from ONE verified k-neighbour search, k = the deepest any of the requested
    analyses needs (rows sorted by distance, as the reference's classes build them: common_neighbor_analysis.py:125-140,
    centro_symmetry_parameter.py:79-100, polyhedral_template_matching.py:110-150).  ``types``: optional int32 per OWNED atom
    (PTM alloy ordering) — it travels with the halo as one more packed column.  Returns (dom, results) with results[name] in
    ``dom`` order; only rows with ``dom.owned`` are meaningful."""
    t = _torch()
    flags = ptm_structure.replace("all", "dcub-dhex-graphene")
    two_shell = "ptm" in what and any(s in flags for s in ("dcub", "dhex", "graphene"))
    k = max(18 if "ptm" in what else 0, 14 if "acna" in what else 0, int(csp_neighbors) if "csp" in what else 0)
    if k <= 0:
        raise ValueError(f"nothing to do: what={what!r}")
    dom, idx, dst, valid = knn_step(dec, x, y, z, gid, k, neighbor_rows=13 if two_shell else 0,
                                    extra=() if types is None else (types,))
    n = int(dom.x.shape[0])
    b = dec.box
    dev = dom.x.device
    where = (dom.x, dom.y, dom.z, b.box, b.origin, b.boundary)
    out = {}
    if "acna" in what:
        out["acna"] = t.zeros((n,), dtype=t.int32, device=dev)
        kernels.cna.acna(*where, idx, out["acna"], 1)
    if "csp" in what:
        out["csp"] = t.zeros((n,), dtype=t.float64, device=dev)
        kernels.csp.get_csp(*where, idx, csp_neighbors, out["csp"], 1)
    if "ptm" in what:
        res = t.zeros((n, 8), dtype=t.float64, device=dev)
        ind = t.zeros((n, 18), dtype=t.int32, device=dev)
        rows18 = idx if k == 18 else idx[:, :18].contiguous()  # PTM reads exactly 18 columns
        ty = dom.extra[0].to(t.int32).contiguous() if types is not None else None
        kernels.ptm.get_ptm(ptm_structure, *where, rows18, ty, ptm_threshold, res, ind, 1)
        out["ptm"], out["ptm_indices"] = res, ind
    out["knn_idx"], out["knn_dist"], out["valid"] = idx, dst, valid
    return dom, out


def steinhardt_step(dec: SlabDecomposition, x, y, z, gid, llist, rc: float, max_neigh: int, average: bool = False,
                    wl: bool = False, wlhat: bool = False):
    """Steinhardt q_l (w_l, w_l-hat) of the owned atoms over all neighbours within ``rc`` (steinhardt_bond_orientation.py:
    cutoff mode, src/steinhardt_bond_orientation.cpp:288-575).  q_lm of an atom needs its neighbours within rc: halo rc.
    ``average=True`` (Lechner-Dellago) also needs the q_lm of those neighbours, i.e. THEIR neighbourhoods: halo 2 rc, no second
    exchange.  Rows come from the keyed neighbor build, so sums run in the order of the undivided system: identical bits.
    Returns (dom, qnarray) in ``dom`` order; rows with ``dom.owned`` are meaningful."""
    t = _torch()
    dom = dec.exchange_halo(x, y, z, gid, (2.0 if average else 1.0) * rc, sort=False)
    n = int(dom.x.shape[0])
    b = dec.box
    dev = dom.x.device
    verlet = t.empty((n, max_neigh), dtype=t.int32, device=dev)
    dist = t.empty((n, max_neigh), dtype=t.float64, device=dev)
    nn = t.empty((n,), dtype=t.int32, device=dev)
    dec.hint_window((2.0 if average else 1.0) * rc, dom.x)
    kernels.neighbor.build_neighbor(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, rc, verlet, dist, nn, 1, fill_pads=True,
                             key=dom.gid if dec.world > 1 else None)
    ll = np.ascontiguousarray(np.asarray(llist), dtype=np.int32)
    nl, lmax = int(ll.shape[0]), int(ll.max())
    qlm_r = t.zeros((n, nl, 2 * lmax + 1), dtype=t.float64, device=dev)
    qlm_i = t.zeros((n, nl, 2 * lmax + 1), dtype=t.float64, device=dev)
    qn = t.zeros((n, nl * (1 + int(bool(wl)) + int(bool(wlhat)))), dtype=t.float64, device=dev)
    kernels.sbo.get_sq(dom.x, dom.y, dom.z, b.box, b.origin, b.boundary, verlet, dist, nn, np.zeros((2, 2)), ll, 0, lmax, wl, wlhat,
                average, False, rc, False, qlm_r, qlm_i, qn, 1)
    return dom, qn


# ------------------------------------------------------------------------------------------------------------------
# reductions over the neighbor list: g(r) and the Warren-Cowley matrix (integer counts, one all-reduce each)
# ------------------------------------------------------------------------------------------------------------------
def rdf_counts_step(dec: SlabDecomposition, dom: LocalDomain, verlet, dist, nn, types, ntype: int, rc: float, nbin: int):
    """Pair counts (Nt,Nt,nbin) of the WHOLE system from every rank's owned rows (`_rdf._rdf`,
    radial_distribution_function.cpp:22-54): ghost rows are switched off through their neighbour count, the integer
    counts (exact in f64 below 2^53) are summed over the ranks.  `types` int32 0-based in ``dom`` order."""
    t = _torch()
    nn_own = t.where(dom.owned, nn, t.zeros_like(nn)).contiguous()
    g = t.zeros((ntype, ntype, nbin), dtype=t.float64, device=dom.x.device)
    kernels.rdf._rdf(verlet, dist, nn_own, types, g, rc, nbin)
    return dec.all_reduce_(g)


def rdf_streaming_step(dec: SlabDecomposition, x, y, z, gid, types, ntype: int, rc: float, nbin: int):
    """Pair counts (Nt,Nt,nbin) of the WHOLE system from the streaming kernel (`_rdf._rdf_streaming`,
    radial_distribution_function.cpp:143-317 — the practical one at cutoffs where a list would be a hundred columns wide):
    owned centres x owned + ghost candidates.  One halo of rc carries the ghosts and their types; the ghosts are given the
    species codes Nt .. 2 Nt - 1, the kernel runs once with 2 Nt species, and of its (2 Nt, 2 Nt, nbin) integer counts the rows
    of owned centres are kept and the two column halves added — no change to the kernel, every ordered pair with an owned
    centre counted exactly once in the whole job; one all-reduce of the integers (exact in f64 below 2^53).
    ``types`` int32 0-based per OWNED atom.  Returns the counts; the normalisation is the caller's, as for the list kernels."""
    t = _torch()
    if dec.world == 1:
        g = t.zeros((ntype, ntype, nbin), dtype=t.float64, device=x.device)
        b = dec.box
        kernels.rdf._rdf_streaming(x, y, z, types.to(t.int32).contiguous(), b.box, b.origin, b.boundary, g, rc, nbin)
        return g
    dom = dec.exchange_halo(x, y, z, gid, rc, sort=False, extra=(types,))
    b = dec.box
    ty2 = (dom.extra[0].to(t.int32) + t.where(dom.owned, 0, ntype).to(t.int32)).contiguous()
    g2 = t.zeros((2 * ntype, 2 * ntype, nbin), dtype=t.float64, device=dom.x.device)
    kernels.rdf._rdf_streaming(dom.x, dom.y, dom.z, ty2, b.box, b.origin, b.boundary, g2, rc, nbin)
    g = (g2[:ntype, :ntype] + g2[:ntype, ntype:]).contiguous()
    return dec.all_reduce_(g)


def wcp_step(dec: SlabDecomposition, dom: LocalDomain, verlet, nn, types, ntype: int):
    """Warren-Cowley matrix of the whole system: Z_mn / Z_m / atoms-per-type counted over owned rows
    (`mdh_wcp_counts`), one int64 all-reduce, then warren_cowley_parameter.cpp:57-75."""
    t = _torch()
    counts = t.zeros((ntype * ntype + 2 * ntype,), dtype=t.int64, device=dom.x.device)
    kernels.wcp.get_wcp_counts(verlet, nn, types, ntype, counts, rows=dom.owned.to(t.uint8).contiguous())
    c = dec.all_reduce_(counts).cpu().numpy().astype(np.int64)
    T = ntype
    zmn, zm, cnt = c[: T * T].reshape(T, T), c[T * T: T * T + T], c[T * T + T:]
    ntot = float(cnt.sum())
    wcp = np.zeros((T, T))
    for a in range(T):
        for bb in range(T):
            conc = float(cnt[bb]) / ntot
            if conc > 0 and zm[a] > 0:
                wcp[a, bb] = 1.0 - float(zmn[a, bb]) / (conc * float(zm[a]))
    return wcp
