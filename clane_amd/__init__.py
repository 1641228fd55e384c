"""clane_amd -- CLANE's iterative embedding loop (``Z <- X + gamma * P @ Z``) on MI355X.

Drop-in for the hot path of helloybz/CLANE (``clane.graph`` / ``clane.similarity`` /
``clane.embedder`` / ``python -m clane``): same Python surface, arithmetic in hand-written
gfx950 HIP kernels behind a C ABI (``include/clane_hip.h``).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
