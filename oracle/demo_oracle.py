"""CPU oracle (TEST INFRASTRUCTURE, not product code).

A plain PyTorch fp32 restatement of the reference's PIFE -> HDM -> ATMoE hot
path, written functionally over a state-dict that uses the reference's own
parameter names.  Each function cites the reference lines it follows
(paths relative to /root/reference).  Pinned against the reference itself:

  * tests/golden/*.npz  -- inputs/outputs produced by importing the reference's
    own modules on CPU (generator: oracle/gen_golden.py, runs only where
    /root/reference exists);
  * tests/test_oracle_vs_reference.py -- full-size checks against the live
    reference when /root/reference is present.

The *glue* (wiring of backbone -> PIFE -> GeneralFusion -> heads and the
return tuple) has no runnable counterpart in this fork of the reference
(SURVEY.md section 0, fact 2); it follows SURVEY.md section 3.2 and is
"documented, unpinned".
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

HDM_BRANCHES = ("r", "n", "t", "rn", "rt", "nt", "rnt")
# which modality streams (0=RGB,1=NI,2=TI) each HDM branch attends over,
# in concatenation order (modeling/moe/AttnMOE.py:163-169)
HDM_STREAMS = {"r": (0,), "n": (1,), "t": (2,), "rn": (0, 1), "rt": (0, 2),
               "nt": (1, 2), "rnt": (0, 1, 2)}


@dataclass
class Spec:
    """Shape/config description of one model instance (mirrors cfg fields)."""
    arch: str = "clip"            # 'clip' (ViT-B-16, configs/*/DeMo.yml:2) or 'vit' (vit_base_patch16_224)
    img_h: int = 256
    img_w: int = 128
    patch: int = 16
    stride: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    feat_dim: int = 512           # 512 for CLIP (visual.proj), 768 for ViT (make_model.py:26-29)
    moe_head: int = 4             # cfg.MODEL.HEAD
    num_classes: int = 171
    camera_num: int = 4
    sie_camera: bool = True
    sie_coe: float = 1.0
    global_local: bool = True
    hdm: bool = True
    atm: bool = True
    direct: int = 1
    miss: str = "nothing"
    hdm_dropout: float = 0.1      # AttnMOE.py:125 default, make_model.py:132 does not override
    use_dgaf: bool = False        # cfg.MODEL.USE_DGAF with DGAF_VERSION 'v3', no SDTPS/FRCA (make_model.py:835-865, "branch 3")
    dgaf_heads: int = 8           # cfg.MODEL.DGAF_NUM_HEADS (config/defaults.py:62)
    dgaf_tau: float = 1.0         # cfg.MODEL.DGAF_TAU (config/defaults.py:60)
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def n_patches(self) -> int:
        return ((self.img_h - self.patch) // self.stride + 1) * ((self.img_w - self.patch) // self.stride + 1)

    @property
    def ln_eps(self) -> float:
        # clip/model.py:154 LayerNorm default eps=1e-5 ; vit_pytorch.py:491 eps=1e-6
        return 1e-5 if self.arch == "clip" else 1e-6

    @property
    def hdm_heads(self) -> int:
        return self.feat_dim // 64   # AttnMOE.py:142


# --------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------
def quick_gelu(x: Tensor) -> Tensor:
    """x * sigmoid(1.702 x)  (AttnMOE.py:5-7, clip/model.py:163-165)."""
    return x * torch.sigmoid(1.702 * x)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def batch_norm(x: Tensor, sd: SD, prefix: str, training: bool, spec: Spec,
               updates: Optional[SD] = None) -> Tensor:
    """nn.BatchNorm1d over (B, F).  Train: batch statistics (biased var for the
    normalisation, unbiased for running_var, momentum 0.1); eval: running stats."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        n = x.shape[0]
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        if updates is not None:
            m = spec.bn_momentum
            updates[prefix + ".running_mean"] = (1 - m) * sd[prefix + ".running_mean"] + m * mean.detach()
            updates[prefix + ".running_var"] = (1 - m) * sd[prefix + ".running_var"] + m * (var.detach() * n / (n - 1))
            updates[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - mean) / torch.sqrt(var + spec.bn_eps) * w + b


def mha_self(x: Tensor, in_w: Tensor, in_b: Optional[Tensor], out_w: Tensor, out_b: Tensor, heads: int) -> Tensor:
    """Multi-head self attention over x (B, L, D); semantics of nn.MultiheadAttention
    (clip/model.py:172,223-225) and of vit_pytorch.py:159-170 (same math)."""
    B, L, D = x.shape
    dh = D // heads
    qkv = linear(x, in_w, in_b).reshape(B, L, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    att = att.softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(B, L, D)
    return linear(o, out_w, out_b)


# --------------------------------------------------------------------------
# backbones
# --------------------------------------------------------------------------
def patchify(img: Tensor, spec: Spec) -> Tensor:
    """(B,3,H,W) -> (B, n_patches, 3*p*p) in conv-weight order (c, ky, kx)."""
    cols = F.unfold(img, kernel_size=spec.patch, stride=spec.stride)   # (B, 3*p*p, n)
    return cols.transpose(1, 2)


def clip_visual(sd: SD, p: str, img: Tensor, cv_emb: Optional[Tensor], spec: Spec,
                taps: Optional[dict] = None) -> Tensor:
    """CLIP VisionTransformer.forward (clip/model.py:446-484) with forward_ori blocks
    (clip/model.py:227-230).  Returns (B, 1+n, feat_dim)."""
    B = img.shape[0]
    w = sd[p + "conv1.weight"]
    x = patchify(img, spec) @ w.reshape(w.shape[0], -1).t()          # conv1, no bias (:432,447)
    cls = sd[p + "class_embedding"].expand(B, 1, -1)                 # :450-452
    x = torch.cat([cls, x], 1)
    if cv_emb is not None:                                           # :453-454 only the cls token
        x = torch.cat([x[:, :1] + cv_emb.reshape(B, 1, -1), x[:, 1:]], 1)
    x = x + sd[p + "positional_embedding"]                           # :455
    x = layer_norm(x, sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], spec.ln_eps)  # :457
    if taps is not None:
        taps["ln_pre"] = x
    for i in range(spec.layers):
        q = f"{p}transformer.resblocks.{i}."
        h = layer_norm(x, sd[q + "ln_1.weight"], sd[q + "ln_1.bias"], spec.ln_eps)
        x = x + mha_self(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"],
                         sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"], spec.heads)
        h = layer_norm(x, sd[q + "ln_2.weight"], sd[q + "ln_2.bias"], spec.ln_eps)
        h = quick_gelu(linear(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"]))
        x = x + linear(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"])
        if taps is not None:
            taps[f"block{i}"] = x
    x = layer_norm(x, sd[p + "ln_post.weight"], sd[p + "ln_post.bias"], spec.ln_eps)  # :481
    return x @ sd[p + "proj"]                                        # :482-484


def vit_visual(sd: SD, p: str, img: Tensor, cam_label: Optional[Tensor], spec: Spec,
               taps: Optional[dict] = None, drop_scales=None) -> Tensor:
    """ImageNet ViT-B `Trans.forward_features` in eval mode (vit_pytorch.py:415-434):
    conv patch-embed with bias, cls + pos + SIE on *all* tokens (:423-424), 12 pre-LN
    blocks with erf-GELU (:173-190,125-141), final LayerNorm eps 1e-6.  Dropout is identity at the rates the configs use
    (0).  DropPath (:48-65,186-187) is identity in eval; in training `drop_scales[i] = (s_attn (B,), s_mlp (B,))` holds block i's
    per-sample factors floor(keep + u) / keep in {0, 1/keep} (None = kept), so that a given mask can be replayed."""
    B = img.shape[0]
    w = sd[p + "patch_embed.proj.weight"]
    x = patchify(img, spec) @ w.reshape(w.shape[0], -1).t() + sd[p + "patch_embed.proj.bias"]
    x = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), x], 1)
    x = x + sd[p + "pos_embed"]
    if spec.sie_camera and spec.camera_num > 1 and cam_label is not None:
        x = x + spec.sie_coe * sd[p + "sie_embed"][cam_label]        # (B,1,D) broadcast over tokens
    for i in range(spec.layers):
        q = f"{p}blocks.{i}."
        sa, sm = drop_scales[i] if drop_scales is not None else (None, None)
        h = layer_norm(x, sd[q + "norm1.weight"], sd[q + "norm1.bias"], spec.ln_eps)
        br = mha_self(h, sd[q + "attn.qkv.weight"], sd[q + "attn.qkv.bias"],
                      sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"], spec.heads)
        x = x + (br if sa is None else br * sa[:, None, None])
        h = layer_norm(x, sd[q + "norm2.weight"], sd[q + "norm2.bias"], spec.ln_eps)
        h = F.gelu(linear(h, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]))
        br = linear(h, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
        x = x + (br if sm is None else br * sm[:, None, None])
        if taps is not None:
            taps[f"block{i}"] = x
    return layer_norm(x, sd[p + "norm.weight"], sd[p + "norm.bias"], spec.ln_eps)


def backbone(sd: SD, img: Tensor, cam_label: Optional[Tensor], spec: Spec, drop_scales=None) -> Tuple[Tensor, Tensor]:
    """build_transformer.forward (meta_arch.py:91-103): -> (patch tokens (B,n,C), cls (B,C))."""
    if spec.arch == "clip":
        cv = None
        if spec.sie_camera:
            cv = spec.sie_coe * sd["BACKBONE.cv_embed"][cam_label]    # :95-97
        x = clip_visual(sd, "BACKBONE.base.", img, cv, spec)
    else:
        x = vit_visual(sd, "BACKBONE.base.", img, cam_label, spec, drop_scales=drop_scales)
    return x[:, 1:], x[:, 0]


# --------------------------------------------------------------------------
# PIFE reduce, HDM, ATMoE, heads
# --------------------------------------------------------------------------
def pife_reduce(sd: SD, p: str, cls: Tensor, patches: Tensor, spec: Spec) -> Tensor:
    """make_model.py:52-58,301-303: reduce(cat[cls, mean_patches]) ; LN(2C) -> Linear -> QuickGELU."""
    z = torch.cat([cls, patches.mean(1)], -1)
    z = layer_norm(z, sd[p + ".0.weight"], sd[p + ".0.bias"], 1e-5)
    return quick_gelu(linear(z, sd[p + ".1.weight"], sd[p + ".1.bias"]))


def hdm(sd: SD, p: str, caches: Sequence[Tensor], globs: Sequence[Tensor], spec: Spec,
        drop_masks: Optional[Dict[str, Tensor]] = None) -> List[Tensor]:
    """GeneralFusion.forward_HDM (AttnMOE.py:153-191): seven single-query
    multi-head cross-attentions over [global; patches] streams.
    `drop_masks[branch]` (B, heads, Lk) of {0,1} reproduces train-mode dropout on the
    attention probabilities (p=0.1, scaled by 1/(1-p)); None = no dropout (eval)."""
    C, H = spec.feat_dim, spec.hdm_heads
    dh = C // H
    streams = [torch.cat([g[:, None], c], 1) for g, c in zip(globs, caches)]   # (B,1+n,C)  :155-165
    outs = []
    for br in HDM_BRANCHES:
        kv = torch.cat([streams[i] for i in HDM_STREAMS[br]], 1)                # :166-169
        B, Lk, _ = kv.shape
        w, b = sd[f"{p}{br}.in_proj_weight"], sd[f"{p}{br}.in_proj_bias"]
        tok = sd[f"{p}{br}_token"].reshape(1, C)
        q = linear(tok, w[:C], b[:C]).reshape(H, dh) * (dh ** -0.5)            # same for every sample
        k = linear(kv, w[C:2 * C], b[C:2 * C]).reshape(B, Lk, H, dh)
        v = linear(kv, w[2 * C:], b[2 * C:]).reshape(B, Lk, H, dh)
        att = torch.einsum("hd,blhd->bhl", q, k).softmax(-1)
        if drop_masks is not None and br in drop_masks:
            att = att * drop_masks[br] / (1.0 - spec.hdm_dropout)
        o = torch.einsum("bhl,blhd->bhd", att, v).reshape(B, C)
        outs.append(linear(o, sd[f"{p}{br}.out_proj.weight"], sd[f"{p}{br}.out_proj.bias"]))
    return outs


def atmoe_gate(sd: SD, p: str, feats: Sequence[Tensor], spec: Spec, training: bool,
               updates: Optional[SD] = None) -> Tensor:
    """CrossAttention gate (AttnMOE.py:50-68): -> (B, H, 7)."""
    C, H = spec.feat_dim, spec.moe_head
    d = C // H
    g = p + "gating_network.gate."
    x = torch.cat(list(feats), -1)
    x = quick_gelu(linear(x, sd[g + "linear_re.0.weight"], sd[g + "linear_re.0.bias"]))
    x = batch_norm(x, sd, g + "linear_re.2", training, spec, updates)
    B = x.shape[0]
    q = linear(x, sd[g + "q_.weight"]).reshape(B, H, d)
    k = linear(torch.stack(list(feats), 1), sd[g + "k_.weight"]).reshape(B, 7, H, d)
    return (torch.einsum("bhd,bjhd->bhj", q, k) * (d ** -0.5)).softmax(-1)


def atmoe(sd: SD, p: str, feats: Sequence[Tensor], spec: Spec, training: bool,
          updates: Optional[SD] = None) -> Tensor:
    """MoM.forward (AttnMOE.py:94-121) -> (B, 7C) with layout j*C + h*d + k."""
    C, H = spec.feat_dim, spec.moe_head
    d = C // H
    gates = atmoe_gate(sd, p, feats, spec, training, updates)
    B = feats[0].shape[0]
    out = feats[0].new_zeros(B, 7, H, d)
    cols = []
    for j in range(7):
        heads = []
        for h in range(H):
            e = f"{p}experts.{h}.expertHead.{j}.mlp."
            y = quick_gelu(linear(feats[j][:, h * d:(h + 1) * d], sd[e + "0.weight"], sd[e + "0.bias"]))
            y = batch_norm(y, sd, e + "2", training, spec, updates)
            heads.append(y * gates[:, h, j, None])
        cols.append(torch.cat(heads, -1))
    del out
    return torch.cat(cols, -1)


def general_fusion(sd: SD, caches, globs, spec: Spec, training: bool, updates: Optional[SD] = None,
                   drop_masks=None) -> Tensor:
    """GeneralFusion.forward (AttnMOE.py:202-224) feature output (the aux loss is the int 0)."""
    feats = hdm(sd, "generalFusion.", caches, globs, spec, drop_masks)
    if spec.atm:
        return atmoe(sd, "generalFusion.moe.", feats, spec, training, updates)
    return torch.cat(feats, -1)


def dgaf_v3(sd: SD, p: str, caches: Sequence[Tensor], heads: int, tau: float, taps: Optional[dict] = None) -> Tensor:
    """DualGatedAdaptiveFusionV3.forward with output_dim = 3C (modeling/dual_gated_fusion.py:558-745):
    attention pooling of each modality's patch tokens by a learned query through ONE shared nn.MultiheadAttention
    (:661-679, no dropout) + LayerNorm, entropy gate IEG (:681-719), importance gate MIG (:721-730), sigmoid(alpha)
    blend (:732-734), LayerNorm(Linear) enhancement added to every pooled modality feature, concatenated (:739-745)."""
    C = caches[0].shape[-1]
    dh = C // heads
    w, b = sd[p + "attn_pool.in_proj_weight"], sd[p + "attn_pool.in_proj_bias"]
    hs = []
    for name, x in zip(("rgb_query", "nir_query", "tir_query"), caches):
        B, N, _ = x.shape
        q = linear(sd[p + name].reshape(1, C), w[:C], b[:C]).reshape(heads, dh) * (dh ** -0.5)
        k = linear(x, w[C:2 * C], b[C:2 * C]).reshape(B, N, heads, dh)
        v = linear(x, w[2 * C:], b[2 * C:]).reshape(B, N, heads, dh)
        att = torch.einsum("hd,bnhd->bhn", q, k).softmax(-1)
        o = torch.einsum("bhn,bnhd->bhd", att, v).reshape(B, C)
        o = linear(o, sd[p + "attn_pool.out_proj.weight"], sd[p + "attn_pool.out_proj.bias"])
        hs.append(layer_norm(o, sd[p + "attn_norm.weight"], sd[p + "attn_norm.bias"], 1e-5))          # :677
    scores = []
    for h in hs:
        a = h.abs() + 1e-8                                                                            # :683-686
        prob = a / a.sum(-1, keepdim=True)
        ent = -(prob * torch.log(prob + 1e-8)).sum(-1)
        z = linear(h, sd[p + "entropy_proj.weight"], sd[p + "entropy_proj.bias"]).mean(-1)            # :705-707
        scores.append(z * torch.exp(-ent / tau))                                                      # :709-711
    ew = torch.stack(scores, -1).softmax(-1)                                                          # :713-714
    h_entropy = sum(ew[:, i:i + 1] * hs[i] for i in range(3))
    g = linear(torch.cat(hs, -1), sd[p + "gate_net.0.weight"], sd[p + "gate_net.0.bias"])             # :609-615
    g = torch.relu(layer_norm(g, sd[p + "gate_net.1.weight"], sd[p + "gate_net.1.bias"], 1e-5))
    gates = torch.sigmoid(linear(g, sd[p + "gate_net.3.weight"], sd[p + "gate_net.3.bias"]))
    h_importance = sum(gates[:, i:i + 1] * hs[i] for i in range(3))
    alpha = torch.sigmoid(sd[p + "_alpha"])                                                           # :640-643
    fused = alpha * h_entropy + (1 - alpha) * h_importance
    enh = layer_norm(linear(fused, sd[p + "modal_enhance.0.weight"], sd[p + "modal_enhance.0.bias"]),
                     sd[p + "modal_enhance.1.weight"], sd[p + "modal_enhance.1.bias"], 1e-5)
    if taps is not None:
        taps.update(dgaf_h=hs, dgaf_ew=ew, dgaf_gates=gates, dgaf_fused=fused, dgaf_enh=enh)
    return torch.cat([h + enh for h in hs], -1)


def bnneck_head(sd: SD, bn: str, cls: str, x: Tensor, spec: Spec, training: bool,
                updates: Optional[SD] = None) -> Tensor:
    """classifier(bottleneck(x)) (make_model.py:134-138,174-194)."""
    return linear(batch_norm(x, sd, bn, training, spec, updates), sd[cls + ".weight"])


def model_forward(sd: SD, spec: Spec, x: Dict[str, Tensor], cam_label: Optional[Tensor] = None,
                  training: bool = False, return_pattern: int = 3, updates: Optional[SD] = None,
                  drop_masks=None, taps: Optional[dict] = None, drop_path_scales=None):
    """Whole hot path with the wiring of SURVEY.md section 3.2.
    drop_path_scales (ViT trunk, training): per block (s_attn, s_mlp), each (3B,) in the order [RGB | NI | TI] (or None)."""
    imgs = [x["RGB"], x["NI"], x["TI"]]
    if not training:                                                  # make_model.py:233-239
        miss = {"r": (0,), "n": (1,), "t": (2,), "rn": (0, 1), "rt": (0, 2), "nt": (1, 2)}.get(spec.miss, ())
        imgs = [torch.zeros_like(im) if i in miss else im for i, im in enumerate(imgs)]
    if "cam_label" in x:
        cam_label = x["cam_label"]
    caches, globs = [], []
    for mi, im in enumerate(imgs):                                    # :247-249 shared backbone
        B_ = im.shape[0]
        ds = None
        if drop_path_scales is not None:
            ds = [tuple(None if t is None else t[mi * B_:(mi + 1) * B_] for t in pair) for pair in drop_path_scales]
        c, g = backbone(sd, im, cam_label, spec, ds)
        caches.append(c)
        globs.append(g)
    if spec.use_dgaf:
        # class DeMo, "branch 3" (make_model.py:835-865): DGAF-V3 on the raw patch tokens; the globals are the raw cls rows
        feat = dgaf_v3(sd, "dgaf.", caches, spec.dgaf_heads, spec.dgaf_tau, taps)
        if taps is not None:
            taps.update(caches=caches, globs=globs, dgaf_feat=feat)
        if not training:
            return feat
        res = (bnneck_head(sd, "bottleneck_dgaf", "classifier_dgaf", feat, spec, True, updates), feat)
        if not spec.direct:
            for s_, g in zip("rnt", globs):
                res += (bnneck_head(sd, f"bottleneck_{s_}", f"classifier_{s_}", g, spec, True, updates), g)
        return res
    if spec.global_local:                                             # :301-303
        globs = [pife_reduce(sd, n, g, c, spec) for n, g, c in zip(("rgb_reduce", "nir_reduce", "tir_reduce"), globs, caches)]
    ori = torch.cat(globs, -1)                                        # :370
    moe_feat = None
    if spec.hdm or spec.atm:
        moe_feat = general_fusion(sd, caches, globs, spec, training, updates, drop_masks)
    if taps is not None:
        taps.update(caches=caches, globs=globs, ori=ori, moe_feat=moe_feat)
    if training:
        res = ()
        if moe_feat is not None:
            res += (bnneck_head(sd, "bottleneck_moe", "classifier_moe", moe_feat, spec, True, updates), moe_feat)
        if spec.direct:
            res += (bnneck_head(sd, "bottleneck", "classifier", ori, spec, True, updates), ori)
        else:
            for s, g in zip("rnt", globs):
                res += (bnneck_head(sd, f"bottleneck_{s}", f"classifier_{s}", g, spec, True, updates), g)
        if moe_feat is not None:
            res += (0,)                                               # loss_moe = reg_weight(0) * 0
        return res
    if moe_feat is None or return_pattern == 1:
        return ori
    if return_pattern == 2:
        return moe_feat
    return torch.cat([moe_feat, ori], -1)                             # "[moe,ori]" (engine/processor.py:220)


# --------------------------------------------------------------------------
# losses (layers/) and evaluator (utils/metrics.py) -- next rows of SURVEY 8(f)
# --------------------------------------------------------------------------
def xent_label_smooth(logits: Tensor, target: Tensor, eps: float = 0.1) -> Tensor:
    """CrossEntropyLabelSmooth (layers/softmax_loss.py:23-34)."""
    logp = F.log_softmax(logits, 1)
    K = logits.shape[1]
    t = torch.zeros_like(logp).scatter_(1, target[:, None], 1.0)
    t = (1 - eps) * t + eps / K
    return (-t * logp).mean(0).sum()


def triplet_soft_margin(feat: Tensor, labels: Tensor) -> Tensor:
    """TripletLoss() without margin (layers/triplet_loss.py:16-31,51-104,121-135):
    batch-hard mining on euclidean distance, SoftMarginLoss(d_an - d_ap, 1)."""
    sq = (feat ** 2).sum(1, keepdim=True)
    dist = (sq + sq.t() - 2 * feat @ feat.t()).clamp(min=1e-12).sqrt()
    same = labels[:, None] == labels[None, :]
    d_ap = torch.where(same, dist, dist.new_full((), -math.inf)).max(1).values
    d_an = torch.where(~same, dist, dist.new_full((), math.inf)).min(1).values
    return F.softplus(-(d_an - d_ap)).mean()


def reid_loss(score: Tensor, feat: Tensor, target: Tensor, id_w: float, tri_w: float) -> Tensor:
    """make_loss 'softmax_triplet' + label smoothing branch (layers/make_loss.py:37-55)."""
    return id_w * xent_label_smooth(score, target) + tri_w * triplet_soft_margin(feat, target)


def train_loss(outputs, target: Tensor, id_w: float = 0.25, tri_w: float = 1.0) -> Tensor:
    """Pairing rule of engine/processor.py:80-97."""
    n = len(outputs) - (len(outputs) % 2)
    loss = 0
    for i in range(0, n, 2):
        loss = loss + reid_loss(outputs[i], outputs[i + 1], target, id_w, tri_w)
    if len(outputs) % 2 == 1:
        loss = loss + outputs[-1]
    return loss


def eval_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank: int = 50):
    """eval_func (utils/metrics.py:110-169): CMC + mAP with same-pid&same-cam removal."""
    import numpy as np
    num_q, num_g = distmat.shape
    max_rank = min(max_rank, num_g)
    order = np.argsort(distmat, axis=1)
    matches = (g_pids[order] == q_pids[:, None]).astype(np.int32)
    cmcs, aps = [], []
    for qi in range(num_q):
        keep = ~((g_pids[order[qi]] == q_pids[qi]) & (g_camids[order[qi]] == q_camids[qi]))
        m = matches[qi][keep]
        if not m.any():
            continue
        c = m.cumsum()
        c[c > 1] = 1
        cmcs.append(c[:max_rank])
        prec = m.cumsum() / (np.arange(m.shape[0]) + 1.0)
        aps.append((prec * m).sum() / m.sum())
    assert cmcs, "all query identities do not appear in gallery"
    cmc = np.asarray(cmcs, dtype=np.float32).sum(0) / len(cmcs)
    return cmc, float(np.mean(aps))


def r1_map(feats: Tensor, pids, camids, num_query: int, feat_norm: bool = True, max_rank: int = 50):
    """R1_mAP_eval.compute (utils/metrics.py:341-369,395-401)."""
    import numpy as np
    if feat_norm:
        feats = F.normalize(feats, dim=1, p=2)
    qf, gf = feats[:num_query], feats[num_query:]
    dist = (qf ** 2).sum(1, keepdim=True) + (gf ** 2).sum(1, keepdim=True).t() - 2 * qf @ gf.t()
    pids, camids = np.asarray(pids), np.asarray(camids)
    return eval_market1501(dist.numpy(), pids[:num_query], pids[num_query:], camids[:num_query], camids[num_query:], max_rank)


# --------------------------------------------------------------------------
# deterministic parameter factory (shapes = reference state-dict, SURVEY 8(a) note)
# --------------------------------------------------------------------------
def init_state_dict(spec: Spec, seed: int = 0, scale_rand_bn: bool = True) -> SD:
    """Seeded random state-dict with the reference's key names/shapes.  Not the reference's
    initialiser: magnitudes are chosen so that every stage is numerically 'alive'
    (non-trivial BN statistics, biases, running stats) for parity testing."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    def lin(name, o, i, bias=True, std=None):
        sd[name + ".weight"] = rn(o, i, std=std if std is not None else i ** -0.5)
        if bias:
            sd[name + ".bias"] = rn(o, std=0.02)

    def ln(name, n):
        sd[name + ".weight"] = 1.0 + rn(n, std=0.05)
        sd[name + ".bias"] = rn(n, std=0.05)

    def bn(name, n):
        sd[name + ".weight"] = 1.0 + rn(n, std=0.05)
        sd[name + ".bias"] = rn(n, std=0.05)
        sd[name + ".running_mean"] = rn(n, std=0.1) if scale_rand_bn else torch.zeros(n)
        sd[name + ".running_var"] = (1.0 + rn(n, std=0.1)).abs() if scale_rand_bn else torch.ones(n)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    W, C = spec.width, spec.feat_dim
    p = "BACKBONE.base."
    pp = spec.patch * spec.patch * 3
    if spec.arch == "clip":
        sd[p + "conv1.weight"] = rn(W, 3, spec.patch, spec.patch, std=pp ** -0.5)
        sd[p + "class_embedding"] = rn(W, std=W ** -0.5)
        sd[p + "positional_embedding"] = rn(spec.n_patches + 1, W, std=W ** -0.5)
        ln(p + "ln_pre", W)
        for i in range(spec.layers):
            q = f"{p}transformer.resblocks.{i}."
            sd[q + "attn.in_proj_weight"] = rn(3 * W, W, std=W ** -0.5)
            sd[q + "attn.in_proj_bias"] = rn(3 * W, std=0.02)
            lin(q + "attn.out_proj", W, W)
            ln(q + "ln_1", W)
            lin(q + "mlp.c_fc", 4 * W, W)
            lin(q + "mlp.c_proj", W, 4 * W)
            ln(q + "ln_2", W)
        ln(p + "ln_post", W)
        sd[p + "proj"] = rn(W, C, std=W ** -0.5)
        if spec.sie_camera:
            sd["BACKBONE.cv_embed"] = rn(spec.camera_num, 1, W, std=0.5)
    else:
        sd[p + "patch_embed.proj.weight"] = rn(W, 3, spec.patch, spec.patch, std=pp ** -0.5)
        sd[p + "patch_embed.proj.bias"] = rn(W, std=0.02)
        sd[p + "cls_token"] = rn(1, 1, W, std=0.02)
        sd[p + "pos_embed"] = rn(1, spec.n_patches + 1, W, std=0.02)
        if spec.sie_camera and spec.camera_num > 1:
            sd[p + "sie_embed"] = rn(spec.camera_num, 1, W, std=0.02)
        for i in range(spec.layers):
            q = f"{p}blocks.{i}."
            ln(q + "norm1", W)
            lin(q + "attn.qkv", 3 * W, W)
            lin(q + "attn.proj", W, W)
            ln(q + "norm2", W)
            lin(q + "mlp.fc1", 4 * W, W)
            lin(q + "mlp.fc2", W, 4 * W)
        ln(p + "norm", W)
    for n in ("rgb_reduce", "nir_reduce", "tir_reduce"):
        ln(n + ".0", 2 * C)
        lin(n + ".1", C, 2 * C)
    if spec.hdm or spec.atm:
        f = "generalFusion."
        for br in HDM_BRANCHES:
            sd[f"{f}{br}_token"] = rn(1, 1, C, std=1.0)
            sd[f"{f}{br}.in_proj_weight"] = rn(3 * C, C, std=C ** -0.5)
            sd[f"{f}{br}.in_proj_bias"] = rn(3 * C, std=0.02)
            lin(f"{f}{br}.out_proj", C, C)
        if spec.atm:
            d = C // spec.moe_head
            for h in range(spec.moe_head):
                for j in range(7):
                    e = f"{f}moe.experts.{h}.expertHead.{j}.mlp."
                    lin(e + "0", d, d)
                    bn(e + "2", d)
            gk = f + "moe.gating_network.gate."
            lin(gk + "linear_re.0", C, 7 * C)
            bn(gk + "linear_re.2", C)
            lin(gk + "q_", C, C, bias=False)
            lin(gk + "k_", C, C, bias=False)
        lin("classifier_moe", spec.num_classes, 7 * C, bias=False, std=0.02)
        bn("bottleneck_moe", 7 * C)
    if spec.use_dgaf:                                                 # dual_gated_fusion.py:586-637, make_model.py:630-634
        d = "dgaf."
        for n in ("rgb_query", "nir_query", "tir_query"):
            sd[d + n] = rn(1, 1, C, std=1.0)
        sd[d + "attn_pool.in_proj_weight"] = rn(3 * C, C, std=C ** -0.5)
        sd[d + "attn_pool.in_proj_bias"] = rn(3 * C, std=0.02)
        lin(d + "attn_pool.out_proj", C, C)
        ln(d + "attn_norm", C)
        lin(d + "entropy_proj", C, C)
        lin(d + "gate_net.0", C, 3 * C)
        ln(d + "gate_net.1", C)
        lin(d + "gate_net.3", 3, C)
        sd[d + "_alpha"] = torch.tensor(0.3)
        lin(d + "modal_enhance.0", C, C)
        ln(d + "modal_enhance.1", C)
        lin("classifier_dgaf", spec.num_classes, 3 * C, bias=False, std=0.02)
        bn("bottleneck_dgaf", 3 * C)
    if spec.direct:
        lin("classifier", spec.num_classes, 3 * C, bias=False, std=0.02)
        bn("bottleneck", 3 * C)
    else:
        for s in "rnt":
            lin(f"classifier_{s}", spec.num_classes, C, bias=False, std=0.02)
            bn(f"bottleneck_{s}", C)
    return sd
