"""Helpers shared by the GPU parity tests."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from demo2_amd.config import get_defaults  # noqa: E402
from oracle import demo_oracle as O  # noqa: E402


def make_cfg(name="RGBNT201", **model_over):
    """cfg of configs/<name>/DeMo.yml with optional MODEL overrides (incl. test-only TRUNK_* geometry)."""
    c = get_defaults()
    c.merge_from_file(os.path.join(ROOT, "demo2_amd", "configs", name, "DeMo.yml"))
    size = model_over.pop("SIZE", None)
    miss = model_over.pop("MISS", None)
    for k, v in model_over.items():
        c.MODEL[k] = v
    if size is not None:
        c.INPUT.SIZE_TRAIN = list(size)
        c.INPUT.SIZE_TEST = list(size)
    if miss is not None:
        c.TEST.MISS = miss
    return c


def small_cfg(arch="clip", head=4, direct=1, **over):
    kw = dict(TRUNK_WIDTH=128, TRUNK_LAYERS=2, TRUNK_OUT_DIM=128, HEAD=head, DIRECT=direct, SIZE=(64, 32))
    if arch == "vit":
        kw["TRANSFORMER_TYPE"] = "vit_base_patch16_224"
        kw["SIE_COE"] = 3.0
    kw.update(over)
    return make_cfg("RGBNT201", **kw)


def spec_of(model) -> O.Spec:
    g, c = model.geom, model.cfg
    return O.Spec(arch=g.arch, img_h=c.INPUT.SIZE_TRAIN[0], img_w=c.INPUT.SIZE_TRAIN[1], patch=g.patch, stride=g.stride,
                  width=g.width, layers=g.layers, heads=g.heads, feat_dim=g.feat_dim, moe_head=g.moe_head,
                  num_classes=model.num_classes, camera_num=model.camera, sie_camera=bool(c.MODEL.SIE_CAMERA),
                  sie_coe=g.sie_coe, global_local=g.global_local, hdm=bool(model.HDM), atm=bool(model.ATM),
                  direct=int(model.direct), miss=c.TEST.MISS, use_dgaf=bool(getattr(model, "USE_DGAF", False)),
                  dgaf_heads=g.dgaf_heads, dgaf_tau=g.dgaf_tau)


def synth_batch(B, h, w, cams, seed=1234, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    x = {k: torch.randn(B, 3, h, w, generator=g) for k in ("RGB", "NI", "TI")}
    cam = torch.randint(0, cams, (B,), generator=g)
    return {k: v.to(device) for k, v in x.items()}, cam.to(device)


def cpu_sd(model):
    return {k: v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()
            for k, v in model.state_dict().items()}
