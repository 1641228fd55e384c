"""The three collectives of the row-partitioned sweep behind one small interface.

``TorchComm`` is the product implementation: ``torch.distributed`` on a process group (backend "nccl" =
RCCL over xGMI on ROCm, one process per GPU; "gloo" in the CPU tests).  The engine never calls
``torch.distributed`` directly, so tests can also drive several engines inside ONE process (threads sharing a
GPU) through an in-memory implementation of the same interface and check the exchange layouts with the real
HIP kernels on a single card.
"""
from __future__ import annotations

import collections
from typing import List, Optional

import torch


class _Done:
    def wait(self):
        return True


class TorchComm:
    def __init__(self, process_group=None, force_collectives: bool = False):
        """``force_collectives``: issue every collective even in a one-rank group (where each is the identity), and
        let ``SweepEngine`` keep the division it was asked for instead of dropping to the one-GPU plan.  That is how
        the RCCL calls -- API, dtypes, stream hand-over of the async forms -- are exercised on a box with a single
        GPU (tests/test_gpu_scale.py)."""
        import torch.distributed as dist
        self._dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if process_group is not None else 1
        self.rank = dist.get_rank(process_group) if process_group is not None else 0
        self.force = bool(force_collectives) and process_group is not None
        self.calls = collections.Counter()

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        if self.world > 1 or self.force:
            self.calls["all_reduce"] += 1
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.pg)

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out = concat over ranks of inp (inp may be the rank's own slice of out: in-place form)."""
        self.calls["all_gather"] += 1
        w = self._dist.all_gather_into_tensor(out, inp, group=self.pg, async_op=async_op)
        return w if async_op else _Done()

    def all_to_all_rows(self, out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int],
                        async_op: bool = False):
        """Row blocks of `inp` (in_splits[q] rows to rank q) -> row blocks of `out` (out_splits[q] rows from q)."""
        self.calls["all_to_all"] += 1
        w = self._dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                         group=self.pg, async_op=async_op)
        return w if async_op else _Done()

    def share_matrices(self, kernels, mine: list) -> list:
        """`mine`: this rank's DeviceBuffers.  Returns, per rank, that rank's matrices: tensors for the rank's own,
        `PeerMatrix` addresses for the others, mapped into this process through hipIpc (the mappings are kept
        alive by the returned `keep` list, second value)."""
        meta = [(b.export(), b.shape) for b in mine]
        everyone = self.all_gather_object(meta)
        views, keep = [], []
        device = mine[0].tensor.device
        for q, items in enumerate(everyone):
            if q == self.rank:
                views.append([b.tensor for b in mine])
                continue
            opened = [kernels.open_shared_matrix(h, shape, mine[0].dtype, device) for h, shape in items]
            keep.extend(opened)
            views.append([b.tensor for b in opened])
        return views, keep

    def all_gather_object(self, obj) -> list:
        if self.world == 1:
            return [obj]
        out: List[Optional[object]] = [None] * self.world
        self._dist.all_gather_object(out, obj, group=self.pg)
        return out
