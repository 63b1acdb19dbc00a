"""minddet import surface (the reference's top-level package is a docstring skeleton:
minddet/__init__.py:1-3); populated here by the MI355X-native implementation in minddet_amd."""
from minddet_amd import __version__  # noqa: F401
