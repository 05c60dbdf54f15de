"""Path shim: `from modeling import make_model` resolves to the MI355X-native implementation."""
from demo2_amd.modeling import build_model, make_model  # noqa: F401
