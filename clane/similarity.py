"""`clane.similarity` -> `clane_amd.similarity` (import shim)."""
from clane_amd.similarity import *  # noqa: F401,F403
from clane_amd import similarity as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
