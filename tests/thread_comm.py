"""In-memory implementation of clane_amd.comm's interface for W "ranks" running as threads of ONE process
(all on the same device).  Test infrastructure: lets the GPU suite run the row-partitioned engine -- halo /
all-gather layouts, chunking, relabelled CSR, send-buffer packing -- with the real HIP kernels on a single
card.  Collectives are realised with a barrier and plain tensor copies; the device is synchronised around
them (the real backend orders them on streams instead)."""
import threading

import torch


class _Done:
    def wait(self):
        return True


class ThreadWorld:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank: int) -> "ThreadComm":
        return ThreadComm(self, rank)


class ThreadComm:
    def __init__(self, shared: ThreadWorld, rank: int):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.force = False          # clane_amd.comm's interface: a one-rank group that insists on its collectives

    def _exchange(self, value):
        """Everyone deposits `value`; returns the list of all ranks' values (valid until the next collective)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.s.slots[self.rank] = value
        self.s.barrier.wait()
        got = list(self.s.slots)
        return got

    def _finish(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.s.barrier.wait()

    def all_reduce_sum(self, t):
        vals = self._exchange(t.clone())
        total = vals[0].clone()
        for v in vals[1:]:
            total += v
        self._finish()
        t.copy_(total)

    def all_gather_into(self, out, inp, async_op=False):
        vals = self._exchange(inp.clone())
        n = inp.shape[0]
        for q, v in enumerate(vals):
            out[q * n:(q + 1) * n].copy_(v)
        self._finish()
        return _Done()

    def all_to_all_rows(self, out, inp, out_splits, in_splits, async_op=False):
        vals = self._exchange((inp.clone(), list(in_splits)))
        off = 0
        for q, (buf, splits) in enumerate(vals):
            start = sum(splits[:self.rank])
            n = splits[self.rank]
            assert n == out_splits[q], (self.rank, q, n, out_splits[q])
            out[off:off + n].copy_(buf[start:start + n])
            off += n
        self._finish()
        return _Done()

    def share_matrices(self, kernels, mine):
        """Same process: the other ranks' tensors themselves."""
        vals = self._exchange([b.tensor for b in mine])
        self._finish()
        return [list(v) for v in vals], []

    def all_gather_object(self, obj):
        vals = self._exchange(obj)
        self._finish()
        return vals
