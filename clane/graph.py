"""`clane.graph` -> `clane_amd.graph` (import shim)."""
from clane_amd.graph import *  # noqa: F401,F403
from clane_amd import graph as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
