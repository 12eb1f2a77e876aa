"""pyskani_amd — MI355X-native drop-in for pyskani's Database.sketch()/query() hot path.

Mirrors the public names of `pyskani` (src/pyskani/__init__.py:2-16 of the reference).
"""
from .database import Context, Database, Hit, Model, Sketch, default_context

__version__ = "0.1.0"
__author__ = "pyskani_amd authors"
SKANI_VERSION = "0.3.0 (restated; see oracle/README.md)"
__all__ = ["Database", "Hit", "Sketch", "Model", "Context", "default_context", "SKANI_VERSION"]
