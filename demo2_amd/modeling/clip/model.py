"""Parameter container for the CLIP ViT-B/16 visual trunk (reference: modeling/clip/model.py:154-484)
and `build_model`, the CLIP state-dict -> module builder (clip/model.py:647-702, visual side only:
the text tower / LoRA / prompt-adapter branches are outside the hot path).
State-dict names equal the reference's: conv1, class_embedding, positional_embedding, ln_pre,
transformer.resblocks.N.{attn.in_proj_weight,attn.in_proj_bias,attn.out_proj,ln_1,mlp.c_fc,mlp.c_proj,ln_2},
ln_post, proj.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..backbones.vit_pytorch import _Holder, trunc_normal_


class LayerNorm(nn.LayerNorm):
    """fp32 LayerNorm (clip/model.py:154-160); here a parameter container."""


class QuickGELU(_Holder):
    pass


class ResidualAttentionBlock(_Holder):
    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)       # parameters only
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):                                      # clip/model.py:213-221
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class Transformer(_Holder):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class VisionTransformer(_Holder):
    def __init__(self, h_resolution, w_resolution, patch_size, stride_size, width, layers, heads, output_dim, cfg=None):
        super().__init__()
        if cfg is not None and (getattr(cfg.MODEL, "PROMPT", False) or getattr(cfg.MODEL, "ADAPTER", False)):
            raise NotImplementedError("MODEL.PROMPT / MODEL.ADAPTER (MambaPro branches) are outside the DeMo hot path")
        self.h_resolution, self.w_resolution = h_resolution, w_resolution
        self.patch_size = patch_size
        self.stride_size = (stride_size, stride_size) if isinstance(stride_size, int) else tuple(stride_size)
        self.width, self.layers, self.heads, self.output_dim = width, layers, heads, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=self.stride_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(h_resolution * w_resolution + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


def resize_pos_embed(posemb, posemb_new, hight, width, cfg=None):
    """clip/model.py:708-725: (1+gs*gs, D) -> (1+h*w, D), bilinear."""
    posemb_token, posemb_grid = posemb[:1], posemb[1:]
    gs_old = int(math.sqrt(len(posemb_grid)))
    posemb_grid = posemb_grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    posemb_grid = F.interpolate(posemb_grid.float(), size=(hight, width), mode='bilinear')
    posemb_grid = posemb_grid.permute(0, 2, 3, 1).reshape(hight * width, -1)
    return torch.cat([posemb_token.float(), posemb_grid], dim=0)


class CLIP(_Holder):
    """Visual-only stand-in for the reference CLIP class: `.visual` is what DeMo uses (meta_arch.py:71)."""

    def __init__(self, visual):
        super().__init__()
        self.visual = visual


def build_model(cfg, state_dict: dict, h_resolution: int, w_resolution: int, vision_stride_size):
    """clip/model.py:647-702 for ViT checkpoints: infer the visual geometry from the CLIP state-dict, build the
    trunk, resize the positional embedding and load the `visual.*` weights (fp32)."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("only CLIP ViT checkpoints are on the DeMo hot path (RN50 is not)")
    width = state_dict["visual.conv1.weight"].shape[0]
    layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    patch = state_dict["visual.conv1.weight"].shape[-1]
    out_dim = state_dict["visual.proj"].shape[1]
    vt = VisionTransformer(h_resolution, w_resolution, patch, vision_stride_size, width, layers, width // 64, out_dim, cfg)
    sd = {k[len("visual."):]: v.float() for k, v in state_dict.items() if k.startswith("visual.")}
    sd["positional_embedding"] = resize_pos_embed(sd["positional_embedding"], vt.positional_embedding, h_resolution, w_resolution)
    print(vt.load_state_dict(sd, strict=False))
    return CLIP(vt).eval()
