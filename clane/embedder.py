"""`clane.embedder` -> `clane_amd.embedder` (import shim)."""
from clane_amd.embedder import *  # noqa: F401,F403
from clane_amd import embedder as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
