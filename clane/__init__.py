"""Import shim: ``clane.*`` -> ``clane_amd.*`` so code written against helloybz/CLANE runs unchanged
(``from clane.graph import Graph``, ``python -m clane ...``).  No logic lives here."""
