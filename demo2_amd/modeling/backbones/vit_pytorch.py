"""Parameter container for the ImageNet ViT-B/16 trunk (reference: modeling/backbones/vit_pytorch.py).

Only the state-dict layout and initialisation live here (names identical to the reference so
its checkpoints load: cls_token, pos_embed, sie_embed, patch_embed.proj, blocks.N.{norm1,attn.qkv,
attn.proj,norm2,mlp.fc1,mlp.fc2}, norm -- vit_pytorch.py:327-413).  The arithmetic runs in the HIP
kernels driven by demo2_amd/modeling/hip_forward.py; these modules have no forward of their own.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container only: the computation runs in demo2_amd HIP kernels")


class _Attn(_Holder):
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(_Holder):
    def __init__(self, dim, mlp_ratio, qkv_bias, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attn(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed_overlap(_Holder):
    def __init__(self, img_size, patch_size, stride_size, in_chans, embed_dim):
        super().__init__()
        self.num_x = (img_size[1] - patch_size) // stride_size[1] + 1
        self.num_y = (img_size[0] - patch_size) // stride_size[0] + 1
        self.num_patches = self.num_x * self.num_y
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride_size)
        n = patch_size * patch_size * embed_dim
        self.proj.weight.data.normal_(0, math.sqrt(2. / n))          # vit_pytorch.py:304-307


class Trans(_Holder):
    """vit_pytorch.py:327-413 parameter layout."""

    def __init__(self, img_size=(256, 128), patch_size=16, stride_size=16, in_chans=3, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, camera=0, view=0, drop_path_rate=0., sie_xishu=1.0, eps=1e-6,
                 **unused):
        super().__init__()
        if isinstance(stride_size, int):
            stride_size = (stride_size, stride_size)
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.patch_size, self.stride_size = patch_size, tuple(stride_size)
        self.eps = eps
        self.patch_embed = PatchEmbed_overlap(img_size, patch_size, stride_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.cam_num, self.view_num, self.sie_xishu = camera, view, sie_xishu
        if camera > 1 and view > 1:
            self.sie_embed = nn.Parameter(torch.zeros(camera * view, 1, embed_dim))
        elif camera > 1:
            self.sie_embed = nn.Parameter(torch.zeros(camera, 1, embed_dim))
        elif view > 1:
            self.sie_embed = nn.Parameter(torch.zeros(view, 1, embed_dim))
        if hasattr(self, "sie_embed"):
            trunc_normal_(self.sie_embed, std=.02)
        self.drop_path_rates = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]   # :393
        self.blocks = nn.ModuleList([Block(embed_dim, mlp_ratio, qkv_bias, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        trunc_normal_(self.cls_token, std=.02)
        trunc_normal_(self.pos_embed, std=.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):                                           # vit_pytorch.py:403-410
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def load_param(self, model_path):
        """vit_pytorch.py:440-465: tolerant loader with bilinear pos-embed resize."""
        param_dict = torch.load(model_path, map_location='cpu', weights_only=True)
        for key in ('model', 'state_dict'):
            if key in param_dict:
                param_dict = param_dict[key]
        own = self.state_dict()
        for k, v in param_dict.items():
            if 'head' in k or 'dist' in k or k not in own:
                continue
            if 'patch_embed.proj.weight' in k and len(v.shape) < 4:
                O, I, H, W = self.patch_embed.proj.weight.shape
                v = v.reshape(O, -1, H, W)
            elif k == 'pos_embed' and v.shape != self.pos_embed.shape:
                v = resize_pos_embed(v, self.pos_embed, self.patch_embed.num_y, self.patch_embed.num_x)
            if own[k].shape == v.shape:
                own[k].copy_(v)
            else:
                print('shape mismatch for {}: {} vs {}'.format(k, tuple(v.shape), tuple(own[k].shape)))


def resize_pos_embed(posemb, posemb_new, hight, width):
    """vit_pytorch.py:468-484."""
    posemb_token, posemb_grid = posemb[:, :1], posemb[0, 1:]
    gs_old = int(math.sqrt(len(posemb_grid)))
    posemb_grid = posemb_grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    posemb_grid = F.interpolate(posemb_grid, size=(hight, width), mode='bilinear')
    posemb_grid = posemb_grid.permute(0, 2, 3, 1).reshape(1, hight * width, -1)
    return torch.cat([posemb_token, posemb_grid], dim=1)


def vit_base_patch16_224(img_size=(256, 128), stride_size=16, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                         camera=0, view=0, local_feature=False, sie_xishu=1.5, **kwargs):
    """vit_pytorch.py:487-495."""
    kwargs.pop("num_classes", None)
    return Trans(img_size=img_size, patch_size=16, stride_size=stride_size, embed_dim=kwargs.pop("embed_dim", 768),
                 depth=kwargs.pop("depth", 12), num_heads=kwargs.pop("num_heads", 12), mlp_ratio=4, qkv_bias=True, camera=camera, view=view, drop_path_rate=drop_path_rate,
                 sie_xishu=sie_xishu, eps=1e-6, **kwargs)
