"""Parameter container for HDM + ATMoE (reference: modeling/moe/AttnMOE.py:24-149).

State-dict names equal the reference's (SURVEY.md 8(a) note): {r,n,t,rn,rt,nt,rnt}_token,
{branch}.in_proj_weight/in_proj_bias/out_proj.{weight,bias},
moe.experts.H.expertHead.J.mlp.{0,2}.*, moe.gating_network.gate.{linear_re.{0,2},q_,k_}.*.
Construction order (hence default initialisation under a fixed torch seed) follows
AttnMOE.py:135-151.  The arithmetic runs in HIP (demo2_amd/modeling/hip_forward.py).
"""
import torch
import torch.nn as nn

from ..backbones.vit_pytorch import _Holder


class QuickGELU(_Holder):
    pass


class Expert(_Holder):
    def __init__(self, input_dim):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(input_dim, input_dim), QuickGELU(), nn.BatchNorm1d(input_dim))


class ExpertHead(_Holder):
    def __init__(self, input_dim, num_experts):
        super().__init__()
        self.expertHead = nn.ModuleList([Expert(input_dim) for _ in range(num_experts)])


class CrossAttention(_Holder):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.linear_re = nn.Sequential(nn.Linear(7 * dim, dim), QuickGELU(), nn.BatchNorm1d(dim))
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q_ = nn.Linear(dim, dim, bias=False)
        self.k_ = nn.Linear(dim, dim, bias=False)


class GatingNetwork(_Holder):
    def __init__(self, input_dim, head):
        super().__init__()
        self.gate = CrossAttention(input_dim, head)


class MoM(_Holder):
    def __init__(self, input_dim, num_experts, head):
        super().__init__()
        if num_experts != 7:
            raise ValueError("ATMoE mixes exactly the 7 HDM features (AttnMOE.py:94; make_model.py:132)")
        self.head_dim, self.head = input_dim // head, head
        self.experts = nn.ModuleList([ExpertHead(self.head_dim, num_experts) for _ in range(head)])
        self.gating_network = GatingNetwork(input_dim, head)


BRANCHES = ("r", "n", "t", "rn", "rt", "nt", "rnt")


class GeneralFusion(_Holder):
    def __init__(self, feat_dim, num_experts, head, reg_weight=0.1, dropout=0.1, cfg=None):
        super().__init__()
        self.reg_weight, self.feat_dim = reg_weight, feat_dim
        self.HDM, self.ATM = cfg.MODEL.HDM, cfg.MODEL.ATM
        self.head = head
        if self.HDM:
            self.dropout = dropout
            scale = feat_dim ** -0.5
            for b in BRANCHES:
                setattr(self, b + "_token", nn.Parameter(scale * torch.randn(1, 1, feat_dim)))
            self.hdm_heads = feat_dim // 64
            for b in BRANCHES:
                setattr(self, b, nn.MultiheadAttention(embed_dim=feat_dim, num_heads=self.hdm_heads, dropout=dropout))
        if self.ATM:
            self.moe = MoM(input_dim=feat_dim, num_experts=num_experts, head=head)
