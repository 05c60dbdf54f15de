"""Path shim: `from config import cfg` resolves to demo2_amd's yacs-free config."""
from demo2_amd.config import CfgNode, cfg, get_defaults  # noqa: F401
