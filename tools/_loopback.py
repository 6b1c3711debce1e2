"""Loop-back stand-in for the ring exchange of mdapy_amd.distributed on ONE GPU (measurement only): the slab receives what
its two periodic neighbours — identical perfect-lattice slabs — would send: its own outgoing layers shifted by one slab
length along the decomposed axis and by one slab's worth of ids.  Everything above the wire (selection, packing, the one
host read, the ghost append, the windowed cell grid, the keyed build) is the product's code."""
import torch
import torch.distributed as dist


class _Work:
    def wait(self):
        pass


class _Op:
    def __init__(self, op, tensor, peer, group=None):
        self.op, self.tensor = op, tensor


def install(dec, slab_length, n_owned):
    """patch torch.distributed for `dec` (a SlabDecomposition without a process group)"""
    dec._host_staged = lambda: False
    dist.all_reduce = lambda tensor, op=None, group=None: tensor  # identical slabs: every rank's largest layer is this rank's

    shifts = {}

    def fake_batch(ops):
        sends = [o for o in ops if o.op is dist.isend]
        recvs = [o for o in ops if o.op is dist.irecv]
        to_right, to_left = sends[0].tensor, sends[1].tensor      # order in _ring: sends = [to right, to left]
        from_left, from_right = recvs[0].tensor, recvs[1].tensor  # recvs = [from left, from right]
        if to_right.dim() == 1 and to_right.numel() > 16:  # the single-message exchange: [count, x row, y row, z row, (extra,) id row]
            # ONE kernel per direction, as a P2P copy would be one: message + (the neighbour slab's offset in x and in the ids)
            key = (to_right.numel(), str(to_right.device))
            if key not in shifts:
                width = 4
                cap = (to_right.numel() - 1) // width
                sh = torch.zeros_like(to_right)
                sh[1:1 + cap] = slab_length; sh[1 + 3 * cap:1 + 4 * cap] = n_owned
                shifts[key] = sh
            torch.sub(to_right, shifts[key], out=from_left)
            torch.add(to_left, shifts[key], out=from_right)
        elif to_right.dim() == 1:
            from_left.copy_(to_right); from_right.copy_(to_left)
        else:
            a = to_right.clone(); a[0] -= slab_length; a[3] -= n_owned
            b = to_left.clone(); b[0] += slab_length; b[3] += n_owned
            from_left.copy_(a); from_right.copy_(b)
        return [_Work()]

    dist.batch_isend_irecv = fake_batch
    dist.P2POp = _Op
