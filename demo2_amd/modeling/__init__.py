"""`from modeling import make_model` surface of the reference (modeling/__init__.py:1)."""
from .make_model import make_model  # noqa: F401
from .clip.model import build_model  # noqa: F401  (CLIP state-dict -> module builder, clip/model.py:647)
