"""Backbone wrapper (reference: modeling/meta_arch.py:8-115): builds the CLIP / ViT parameter
container, owns the camera embedding `cv_embed`, and provides the BNNeck / classifier initialisers.
Pretrained files are optional here (there is no network in the build environment): a missing
`../ViT-B-16.pt` / PRETRAIN_PATH_T leaves the seeded random initialisation in place.
"""
import os

import torch
import torch.nn as nn

from .backbones.vit_pytorch import _Holder, trunc_normal_
from .clip.model import VisionTransformer, build_model as clip_build_model


def weights_init_kaiming(m):                                       # meta_arch.py:8-22
    classname = m.__class__.__name__
    if classname.find('Linear') != -1:
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out')
        nn.init.constant_(m.bias, 0.0)
    elif classname.find('Conv') != -1:
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)
    elif classname.find('BatchNorm') != -1:
        if m.affine:
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0.0)


def weights_init_classifier(m):                                    # meta_arch.py:25-29
    if m.__class__.__name__.find('Linear') != -1:
        nn.init.normal_(m.weight, std=0.001)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0.0)


CLIP_PRETRAINED = '../ViT-B-16.pt'                                 # make_model_clipreid.py:176


class build_transformer(_Holder):
    def __init__(self, num_classes, cfg, camera_num, view_num, factory, feat_dim):
        super().__init__()
        self.in_planes = feat_dim
        self.cv_embed_sign = cfg.MODEL.SIE_CAMERA
        self.model_name = cfg.MODEL.TRANSFORMER_TYPE
        self.camera_num = camera_num if cfg.MODEL.SIE_CAMERA else 0
        self.view_num = 0
        h, w = cfg.INPUT.SIZE_TRAIN
        sh, sw = cfg.MODEL.STRIDE_SIZE
        if cfg.MODEL.TRANSFORMER_TYPE == 'vit_base_patch16_224':
            self.clip = 0
            self.base = factory[cfg.MODEL.TRANSFORMER_TYPE](
                img_size=cfg.INPUT.SIZE_TRAIN, sie_xishu=cfg.MODEL.SIE_COE, camera=self.camera_num, view=self.view_num,
                stride_size=cfg.MODEL.STRIDE_SIZE, drop_path_rate=cfg.MODEL.DROP_PATH,
                **({'embed_dim': cfg.MODEL.TRUNK_WIDTH, 'depth': cfg.MODEL.TRUNK_LAYERS, 'num_heads': cfg.MODEL.TRUNK_WIDTH // 64}
                   if hasattr(cfg.MODEL, 'TRUNK_WIDTH') else {}))
            path = cfg.MODEL.PRETRAIN_PATH_T
            if path and os.path.isfile(path):
                self.base.load_param(path)
                print('Loading pretrained model from ImageNet')
        elif cfg.MODEL.TRANSFORMER_TYPE == 'ViT-B-16':
            self.clip = 1
            self.sie_xishu = cfg.MODEL.SIE_COE
            width = getattr(cfg.MODEL, 'TRUNK_WIDTH', 768)
            nh, nw = (h - 16) // sh + 1, (w - 16) // sw + 1
            if os.path.isfile(CLIP_PRETRAINED):
                try:
                    sd = torch.jit.load(CLIP_PRETRAINED, map_location="cpu").state_dict()
                except RuntimeError:
                    sd = torch.load(CLIP_PRETRAINED, map_location="cpu", weights_only=True)
                self.base = clip_build_model(cfg, sd, nh, nw, (sh, sw)).visual
                print('Loading pretrained model from CLIP')
            else:
                # TRUNK_* are optional geometry overrides (tests use a narrow trunk); the shipped configs never set them
                self.base = VisionTransformer(nh, nw, 16, (sh, sw), width, getattr(cfg.MODEL, 'TRUNK_LAYERS', 12),
                                              width // 64, feat_dim, cfg)
            if cfg.MODEL.SIE_CAMERA and cfg.MODEL.SIE_VIEW:
                n = camera_num * view_num
            elif cfg.MODEL.SIE_CAMERA:
                n = camera_num
            elif cfg.MODEL.SIE_VIEW:
                n = view_num
            else:
                n = 0
            if n:
                self.cv_embed = nn.Parameter(torch.zeros(n, 1, width))   # meta_arch.py:75-86
                trunc_normal_(self.cv_embed, std=1e-6)
        else:
            raise NotImplementedError(f"backbone {cfg.MODEL.TRANSFORMER_TYPE} is outside the DeMo hot path "
                                      "(SURVEY.md section 2 row 15)")
        self.num_classes = num_classes
