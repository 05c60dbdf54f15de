"""Parameter container for DGAF-V3 (reference: modeling/dual_gated_fusion.py:558-637).

State-dict names equal the reference's: {rgb,nir,tir}_query (1,1,C); attn_pool.{in_proj_weight,in_proj_bias,out_proj.*};
attn_norm.*; entropy_proj.*; gate_net.{0,1,3}.*; _alpha (); modal_enhance.{0,1}.*  -- so a reference checkpoint of the `DeMo`
class with MODEL.USE_DGAF loads through load_param.  The arithmetic runs in HIP (hip_forward.dgaf_forward, csrc/dgaf.hip).
Only the 3-modality, output_dim = 3C form is built: that is what make_model.py:601-617 instantiates without FRCA.
"""
import torch
import torch.nn as nn

from .backbones.vit_pytorch import _Holder


class _Slot(_Holder):
    """Parameter-free position inside an nn.Sequential (ReLU / Sigmoid of the reference), keeps the indices 0,1,3."""


class DualGatedAdaptiveFusionV3(_Holder):
    def __init__(self, feat_dim, output_dim=None, tau=1.0, init_alpha=0.5, num_heads=8):
        super().__init__()
        C = feat_dim
        if (output_dim or 3 * C) != 3 * C:
            raise NotImplementedError("DGAF-V3 with a single fused output (output_dim == feat_dim) is not used by make_model")
        if C % num_heads:
            raise ValueError(f"feat_dim {C} is not divisible by DGAF_NUM_HEADS {num_heads}")
        self.feat_dim, self.output_dim, self.tau, self.num_heads = C, 3 * C, float(tau), num_heads
        for n in ("rgb_query", "nir_query", "tir_query"):
            setattr(self, n, nn.Parameter(C ** -0.5 * torch.randn(1, 1, C)))
        self.attn_pool = nn.MultiheadAttention(embed_dim=C, num_heads=num_heads, batch_first=True)
        self.attn_norm = nn.LayerNorm(C)
        self.entropy_proj = nn.Linear(C, C)
        self.gate_net = nn.Sequential(nn.Linear(3 * C, C), nn.LayerNorm(C), _Slot(), nn.Linear(C, 3), _Slot())
        self._alpha = nn.Parameter(torch.tensor(float(init_alpha)))
        self.modal_enhance = nn.Sequential(nn.Linear(C, C), nn.LayerNorm(C))
