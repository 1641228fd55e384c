"""Similarity plugins -- same names, constructor arguments and call protocol as the
reference's ``clane/similarity.py`` (selected by NAME from this module's namespace by the
CLI, reference ``__main__.py:39-48``).

``CosineSimilarity.__call__`` runs in a HIP kernel (``clane_pair_cosine_*``); inside
``Graph.build_P`` the gather + similarity + softmax are fused on the GPU and this callable is
not invoked at all for ``CosineSimilarity``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _hip


class Similarity:
    """Plugin protocol (reference similarity.py:5-7): a callable ``sim(v1[B,d], v2[B,d]) -> [B]``."""

    def is_trainable(self) -> bool:
        return isinstance(self, nn.Module)


class CosineSimilarity(Similarity):
    """Batched row-wise "cosine" exactly as the reference computes it (similarity.py:26-37):

        out[i] = dot(v1[i], v2[i]) / (||v1||_F * ||v2||_F)

    The two norms are taken over the WHOLE batch (``pow(2).sum()`` has no ``dim``), so this is
    a true cosine only for a single pair.  ``mode="per_edge"`` (an extension, off by default)
    makes ``Graph.build_P`` use true per-edge cosines instead; it does not change ``__call__``.
    Unknown keyword arguments are accepted and ignored, like the reference (``foo: bar`` in
    its tests/config.yaml).
    """

    def __init__(self, **kwargs) -> None:
        super().__init__()
        mode = kwargs.get("mode", "reference")
        if mode not in ("reference", "per_edge"):
            raise ValueError(f"CosineSimilarity mode must be 'reference' or 'per_edge', got {mode!r}")
        self.mode = mode

    def __call__(self, v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
        if v1.dim() == 1:
            v1 = v1.unsqueeze(0)
        if v2.dim() == 1:
            v2 = v2.unsqueeze(0)
        if v1.dim() != 2 or v1.shape != v2.shape:
            raise ValueError(f"CosineSimilarity: expected two [B, d] batches of equal shape, got "
                             f"{tuple(v1.shape)} and {tuple(v2.shape)}")
        dtype = torch.promote_types(v1.dtype, v2.dtype)
        if dtype not in (torch.float32, torch.float64):
            raise TypeError(f"CosineSimilarity: float32 or float64 inputs required, got {v1.dtype}/{v2.dtype}")
        k = _hip.kernels()
        home = v1.device
        dev = _hip.require_gpu(home)
        a = v1.detach().to(dev, dtype).contiguous()
        b = v2.detach().to(dev, dtype).contiguous()
        out = torch.empty(a.shape[0], dtype=dtype, device=dev)
        ws = torch.empty(k.reduce_ws_len(), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            k.pair_cosine(a, b, a.shape[1], out, ws)
        return out.to(home)


class AsymmertricSimilarity(nn.Module, Similarity):
    """Learnable bilinear score (reference similarity.py:40-57; the class name's spelling is API).

    Kept importable for configuration compatibility only: the reference can reach it solely
    through ``IterativeEmbedder``, which fails at construction upstream, so
    the trainable path is outside the hot path this package implements.
    """

    def __init__(self, n_dim: int, **kwargs) -> None:
        super().__init__()
        self.Phi_src = nn.Linear(n_dim, n_dim, bias=False)
        self.Phi_dst = nn.Linear(n_dim, n_dim, bias=False)
        nn.init.xavier_normal_(self.Phi_src.weight)
        nn.init.xavier_normal_(self.Phi_dst.weight)

    def forward(self, z_src: torch.Tensor, z_dst: torch.Tensor) -> torch.Tensor:
        return (self.Phi_src(z_src) * self.Phi_dst(z_dst)).sum(-1)
