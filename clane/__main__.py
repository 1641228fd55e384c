"""`python -m clane ...` -> clane_amd's CLI (import shim)."""
from clane_amd.__main__ import embedding, get_parser, main  # noqa: F401

if __name__ == "__main__":
    main()
