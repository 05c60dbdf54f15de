"""Kernel schedule of the hot path: PIFE ViT trunk -> PIFE reduce -> HDM -> ATMoE -> BNNeck heads.

Everything below is a fixed sequence of launches into libdemo_hip.so (demo2_amd/ops.py); there is no
PyTorch arithmetic on the data path.  The three modalities share one trunk (make_model.py:247-249 of the
reference), so they run as ONE batch of 3B images: exact, because the trunk has LayerNorm only.

Data layout in HBM (B = samples, L = 1 + patches, W = trunk width, C = feat dim, T = compute dtype):
  cols    (3B*np, 3*p*p) T      im2col patches, modality-major [RGB | NI | TI]
  x       (3B*L, W)      fp32   residual stream (kept fp32 in both modes)
  h/qkv/a/u (3B*L, .)    T      LayerNorm outputs, packed q|k|v, attention output, MLP hidden
  tokens  (3B*L, C)      T      trunk output; row 0 of every image is later overwritten by the PIFE global
  ori     (B, 3C)        fp32   [R_global | N_global | T_global]
  feats   (B, 7, C)      fp32   HDM outputs == ATMoE query (B,7C) == ATMoE key stack (B,7,C)
  moe     (B, 7C)        fp32   layout j*C + h*d + k   (AttnMOE.py:116-117)
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops
from ..ops import ACT_GELU, ACT_NONE, ACT_QUICKGELU

Tensor = torch.Tensor
BRANCHES = ("r", "n", "t", "rn", "rt", "nt", "rnt")
NC_PAD = 128            # HDM score columns padded to the MFMA kernel's N granule


class Tape:
    """Activations kept for the backward pass (288 GB of HBM: nothing is recomputed)."""

    def __init__(self):
        self.blocks: List[dict] = []
        self.d: Dict[str, object] = {}


def trunk_forward(m, imgs: List[Tensor], cam: Optional[Tensor], T: torch.dtype, zero_mod=(), tape: Optional[Tape] = None,
                  drop_path: bool = False) -> Tensor:
    """Shared trunk over [RGB; NI; TI].  Returns tokens (3B*L, C) in dtype T.
    CLIP: clip/model.py:446-484 + forward_ori :227-230.  ViT: vit_pytorch.py:415-434,173-190.
    drop_path (training): the ViT blocks' stochastic depth, x + scale[image] * f(x) with scale in {0, 1/keep} per image and branch
    (vit_pytorch.py:48-65,186-187; rates linspace(0, DROP_PATH, depth), :393).  The masks come from the library's counter RNG (they
    cannot match torch's stream; m.last_drop_path keeps them so a test can feed them to the oracle).  Such a block's residual
    add is a separate row-scaling launch instead of the GEMM epilogue: config V only, not the benchmarked path."""
    g = m.geom
    B = imgs[0].shape[0]
    Bn, np_, L, W = 3 * B, g.n_patches, g.n_patches + 1, g.width
    P = m.ptab
    dev = imgs[0].device
    # ---- K1: patch embedding = im2col + GEMM ---------------------------------------------------
    cols = torch.empty(Bn * np_, 3 * g.patch * g.patch, device=dev, dtype=T)
    for i, im in enumerate(imgs):
        sl = cols[i * B * np_:(i + 1) * B * np_]
        if i in zero_mod:
            sl.zero_()                                    # eval-time missing modality (make_model.py:233-239)
        else:
            ops.im2col_into(im, sl, g.patch, g.stride)
    wconv = P.w(g.k_conv, T).view(W, -1)
    patch = ops.gemm(cols, wconv, bias=P.f(g.k_conv_bias) if g.k_conv_bias else None)
    # ---- K2: cls/pos/SIE (+ ln_pre) ----------------------------------------------------------------
    cam3 = cam.repeat(3) if (cam is not None and g.cam_embed) else None
    emb = ops.embed_tokens(patch, P.f(g.k_cls).view(-1), P.f(g.k_pos).view(L, W), P.f(g.cam_embed).view(-1, W) if g.cam_embed else None,
                           cam3, g.sie_coe, g.cam_all_tokens, P.f(g.k_lnpre + ".weight") if g.k_lnpre else None,
                           P.f(g.k_lnpre + ".bias") if g.k_lnpre else None, g.ln_eps, Bn, np_, save=tape is not None,
                           err=m.device_errors() if cam3 is not None else None)
    if tape is not None:
        x, pre, mean, rstd = emb
        tape.d.update(cols=cols, emb_pre=pre, emb_mean=mean, emb_rstd=rstd, cam3=cam3)
    else:
        x = emb
    act = ACT_QUICKGELU if g.arch == "clip" else ACT_GELU
    rates = g.drop_path if (drop_path and g.drop_path) else None
    m.last_drop_path = [] if rates else None
    if T == torch.bfloat16 and m.bf16_residual and m.bf16_grad_stream and g.arch == "clip" and not rates:
        return _trunk_blocks_bf16_residual(m, x, Bn, L, W, act, tape)
    if T == torch.bfloat16 and m.bf16_fwd_stream and m.bf16_grad_stream and g.arch == "clip":
        return _trunk_blocks_bf16_stream(m, x, Bn, L, W, act, tape)
    # ---- K3-K6: transformer blocks -------------------------------------------------------------------
    # bf16 mode: fc1's epilogue saves act'(pre) instead of pre (both from one exp / rcp), so the backward product (dY W2^T) * act'
    # multiplies by a saved tensor instead of re-evaluating the derivative (GEMM epilogue VALU: ~10 -> 3 instructions per element)
    deriv = bool(m.save_act_deriv and T == torch.bfloat16)
    for i in range(g.layers):
        k = g.block_keys(i)
        sv = tape is not None
        dp = rates[i] if rates else 0.0
        dp_a = dp_m = None
        if dp > 0.0:
            dp_a = ops.droppath_mask(Bn, 1.0 - dp, m.next_seed(), dev, m.seed_dev)
            dp_m = ops.droppath_mask(Bn, 1.0 - dp, m.next_seed(), dev, m.seed_dev)
        if rates:
            m.last_drop_path.append((dp_a, dp_m))
        r1 = ops.layernorm_fwd(x, P.f(k["ln1"] + ".weight"), P.f(k["ln1"] + ".bias"), g.ln_eps, T, save=sv)
        h1 = r1[0] if sv else r1
        qkv = ops.gemm(h1, P.w(k["qkv_w"], T), bias=P.f(k["qkv_b"]))
        r2 = ops.attention_fwd(qkv, Bn, L, g.heads, save_lse=sv)
        a = r2[0] if sv else r2
        if dp_a is None:
            x1 = ops.gemm(a, P.w(k["proj_w"], T), bias=P.f(k["proj_b"]), residual=x, out=None if sv else x, out_dtype=torch.float32)
        else:
            x1 = ops.rowscale_add(x, ops.gemm(a, P.w(k["proj_w"], T), bias=P.f(k["proj_b"])), dp_a, Bn, L, out=None if sv else x)
        r3 = ops.layernorm_fwd(x1, P.f(k["ln2"] + ".weight"), P.f(k["ln2"] + ".bias"), g.ln_eps, T, save=sv)
        h2 = r3[0] if sv else r3
        if sv:
            pre = torch.empty(Bn * L, 4 * W, device=dev, dtype=torch.uint8 if (deriv and m.act_deriv_d8) else T)     # uint8: DEMO_D8 codes of act'
            u = ops.gemm(h2, P.w(k["fc1_w"], T), bias=P.f(k["fc1_b"]), act=act, pre_out=pre, pre_out_deriv=deriv)
        else:
            pre = None
            u = ops.gemm(h2, P.w(k["fc1_w"], T), bias=P.f(k["fc1_b"]), act=act)
        if dp_m is None:
            x2 = ops.gemm(u, P.w(k["fc2_w"], T), bias=P.f(k["fc2_b"]), residual=x1, out=None if sv else x1, out_dtype=torch.float32)
        else:
            x2 = ops.rowscale_add(x1, ops.gemm(u, P.w(k["fc2_w"], T), bias=P.f(k["fc2_b"])), dp_m, Bn, L, out=None if sv else x1)
        if sv:
            tape.blocks.append(dict(x=x, mean1=r1[1], rstd1=r1[2], h1=h1, qkv=qkv, lse=r2[1], a=a, x1=x1, mean2=r3[1], rstd2=r3[2],
                                    h2=h2, pre=pre, pre_deriv=deriv, u=u, dp_a=dp_a, dp_m=dp_m))
        x = x2
    # ---- K7: final LayerNorm (+ CLIP projection) ---------------------------------------------------------
    sv = tape is not None
    rf = ops.layernorm_fwd(x, P.f(g.k_lnpost + ".weight"), P.f(g.k_lnpost + ".bias"), g.ln_eps, T, save=sv)
    y = rf[0] if sv else rf
    return _trunk_tail(m, x, y, rf, tape)


def _trunk_tail(m, x_final: Tensor, y: Tensor, rf, tape: Optional[Tape]) -> Tensor:
    g, P = m.geom, m.ptab
    T = y.dtype
    if g.arch == "clip":
        tokens = ops.gemm(y, P.w_t(g.k_proj, T))          # (3B*L, C) = y @ proj ; proj stored (W, C)
    else:
        tokens = y
    if tape is not None:
        tape.d.update(x_final=x_final, meanf=rf[1], rstdf=rf[2], y=y)
    return tokens


def _trunk_blocks_bf16_residual(m, x32: Tensor, Bn: int, L: int, W: int, act: int, tape: Optional[Tape]) -> Tensor:
    """Transformer blocks with the residual stream in bf16 and the adds INSIDE the GEMM epilogues (bf16 compute mode): the output
    projection and fc2 read the stream as a bf16 residual operand and write the new stream in bf16 -- 134 instead of 268 MB of epilogue
    traffic per product at B = 128 (the fp32 form is HBM-bound there: tools/gemm_cost.py, 126 us against 57 us without its epilogue) --
    and the LayerNorms (forward and backward) read a bf16 stream.  One rounding of the stream to bf16 per residual add (24 per trunk)."""
    g, P = m.geom, m.ptab
    T = torch.bfloat16
    dev = x32.device
    sv = tape is not None
    deriv = bool(m.save_act_deriv)
    xs = ops.cast_f32_bf16(x32.view(-1), torch.empty(x32.numel(), device=dev, dtype=T)).view_as(x32)
    for i in range(g.layers):
        k = g.block_keys(i)
        r1 = ops.layernorm_add_fwd(xs, None, P.f(k["ln1"] + ".weight"), P.f(k["ln1"] + ".bias"), g.ln_eps, save=sv)
        h1 = r1[0] if sv else r1
        qkv = ops.gemm(h1, P.w(k["qkv_w"], T), bias=P.f(k["qkv_b"]))
        r2 = ops.attention_fwd(qkv, Bn, L, g.heads, save_lse=sv)
        a = r2[0] if sv else r2
        x1 = ops.gemm(a, P.w(k["proj_w"], T), bias=P.f(k["proj_b"]), residual=xs, out=None if sv else xs, out_dtype=T)
        r3 = ops.layernorm_add_fwd(x1, None, P.f(k["ln2"] + ".weight"), P.f(k["ln2"] + ".bias"), g.ln_eps, save=sv)
        h2 = r3[0] if sv else r3
        if sv:
            pre = torch.empty(Bn * L, 4 * W, device=dev, dtype=torch.uint8 if (deriv and m.act_deriv_d8) else T)     # uint8: DEMO_D8 codes of act'
            u = ops.gemm(h2, P.w(k["fc1_w"], T), bias=P.f(k["fc1_b"]), act=act, pre_out=pre, pre_out_deriv=deriv)
        else:
            pre = None
            u = ops.gemm(h2, P.w(k["fc1_w"], T), bias=P.f(k["fc1_b"]), act=act)
        x2 = ops.gemm(u, P.w(k["fc2_w"], T), bias=P.f(k["fc2_b"]), residual=x1, out=None if sv else x1, out_dtype=T)
        if sv:
            tape.blocks.append(dict(x=xs, mean1=r1[1], rstd1=r1[2], h1=h1, qkv=qkv, lse=r2[1], a=a, x1=x1, mean2=r3[1], rstd2=r3[2],
                                    h2=h2, pre=pre, pre_deriv=deriv, u=u, dp_a=None, dp_m=None))
        xs = x2
    rf = ops.layernorm_add_fwd(xs, None, P.f(g.k_lnpost + ".weight"), P.f(g.k_lnpost + ".bias"), g.ln_eps, save=sv)
    y = rf[0] if sv else rf
    return _trunk_tail(m, xs, y, rf, tape)


def _trunk_blocks_bf16_stream(m, x32: Tensor, Bn: int, L: int, W: int, act: int, tape: Optional[Tape]) -> Tensor:
    """CLIP blocks with the residual stream in bf16 (bf16 compute mode): every residual add is fused into the LayerNorm that follows it
    (demo_layernorm_add_fwd: x + branch, rounded to bf16, saved, normalised in one pass), so the attention projection and fc2 have
    plain bf16 epilogues and nothing reads or writes a fp32 stream: 380 instead of 532 MB per (branch GEMM epilogue + LayerNorm) pair,
    and the LayerNorm backwards re-read 76 instead of 152 MB."""
    g, P = m.geom, m.ptab
    T = torch.bfloat16
    dev = x32.device
    sv = tape is not None
    deriv = bool(m.save_act_deriv)
    xs = ops.cast_f32_bf16(x32.view(-1), torch.empty(x32.numel(), device=dev, dtype=T)).view_as(x32)
    pend = None                                            # branch output still to be added to the stream
    for i in range(g.layers):
        k = g.block_keys(i)
        x_in = torch.empty_like(xs) if (sv and pend is not None) else xs
        r1 = ops.layernorm_add_fwd(xs, pend, P.f(k["ln1"] + ".weight"), P.f(k["ln1"] + ".bias"), g.ln_eps,
                                   xsum_out=x_in if pend is not None else None, save=sv)
        h1 = r1[0] if sv else r1
        qkv = ops.gemm(h1, P.w(k["qkv_w"], T), bias=P.f(k["qkv_b"]))
        r2 = ops.attention_fwd(qkv, Bn, L, g.heads, save_lse=sv)
        a = r2[0] if sv else r2
        t_attn = ops.gemm(a, P.w(k["proj_w"], T), bias=P.f(k["proj_b"]))
        x1 = torch.empty_like(xs) if sv else x_in
        r3 = ops.layernorm_add_fwd(x_in, t_attn, P.f(k["ln2"] + ".weight"), P.f(k["ln2"] + ".bias"), g.ln_eps, xsum_out=x1, save=sv)
        h2 = r3[0] if sv else r3
        if sv:
            pre = torch.empty(Bn * L, 4 * W, device=dev, dtype=torch.uint8 if (deriv and m.act_deriv_d8) else T)     # uint8: DEMO_D8 codes of act'
            u = ops.gemm(h2, P.w(k["fc1_w"], T), bias=P.f(k["fc1_b"]), act=act, pre_out=pre, pre_out_deriv=deriv)
        else:
            pre = None
            u = ops.gemm(h2, P.w(k["fc1_w"], T), bias=P.f(k["fc1_b"]), act=act)
        pend = ops.gemm(u, P.w(k["fc2_w"], T), bias=P.f(k["fc2_b"]))
        if sv:
            tape.blocks.append(dict(x=x_in, mean1=r1[1], rstd1=r1[2], h1=h1, qkv=qkv, lse=r2[1], a=a, x1=x1, mean2=r3[1], rstd2=r3[2],
                                    h2=h2, pre=pre, pre_deriv=deriv, u=u, dp_a=None, dp_m=None))
        xs = x1
    x_final = torch.empty_like(xs) if sv else xs
    rf = ops.layernorm_add_fwd(xs, pend, P.f(g.k_lnpost + ".weight"), P.f(g.k_lnpost + ".bias"), g.ln_eps, xsum_out=x_final, save=sv)
    y = rf[0] if sv else rf
    return _trunk_tail(m, x_final, y, rf, tape)


def pife_forward(m, tokens: Tensor, B: int, tape: Optional[Tape] = None) -> Tensor:
    """PIFE global-local reduce for the three modalities (make_model.py:52-58,301-303) -> ori (B, 3C) fp32.
    Without GLOBAL_LOCAL the globals are the cls rows."""
    g = m.geom
    L, C = g.n_patches + 1, g.feat_dim
    P = m.ptab
    dev = tokens.device
    ori = torch.empty(B, 3 * C, device=dev, dtype=torch.float32)
    if not g.global_local:
        for i in range(3):
            ops.get_token0(tokens[i * B * L:(i + 1) * B * L], ori, i * C, 3 * C, B, L)
        return ori
    z = ops.pife_pool_cat(tokens, 3 * B, L)                                   # (3B, 2C) fp32
    zl = torch.empty(3 * B, 2 * C, device=dev, dtype=torch.float32)
    stats = []
    for i, name in enumerate(("rgb_reduce", "nir_reduce", "tir_reduce")):
        r = ops.layernorm_fwd(z[i * B:(i + 1) * B], P.f(name + ".0.weight"), P.f(name + ".0.bias"), 1e-5, torch.float32,
                              save=tape is not None, out=zl[i * B:(i + 1) * B])
        if tape is not None:
            stats.append((r[1], r[2]))
    # one grouped GEMM for the three reduce layers (their parameters are equally spaced in the flat buffer)
    s_w = P.stride("rgb_reduce.1.weight", "nir_reduce.1.weight", "tir_reduce.1.weight")
    s_b = P.stride("rgb_reduce.1.bias", "nir_reduce.1.bias", "tir_reduce.1.bias")
    pre = torch.empty(B, 3 * C, device=dev, dtype=torch.float32) if tape is not None else None
    ops.gemm_raw(zl, P.flat, ori, B, C, 2 * C, 2 * C, 1, 2 * C, 1, 3 * C, b_off=P.off("rgb_reduce.1.weight"),
                 bias=P.flat, bias_off=P.off("rgb_reduce.1.bias"), act=ACT_QUICKGELU, batch1=3, batch2=1,
                 sA=(B * 2 * C, 0), sB=(s_w, 0), sC=(C, 0), sBias=(s_b, 0), pre_out=pre)
    if tape is not None:
        tape.d.update(pife_z=z, pife_zl=zl, pife_stats=stats, pife_pre=pre)
    return ori


def hdm_forward(m, tokens: Tensor, ori: Tensor, B: int, training: bool, tape: Optional[Tape] = None) -> Tensor:
    """HDM (AttnMOE.py:153-191) in the folded single-query form (see csrc/fusion.hip) -> feats (B,7,C) fp32."""
    g = m.geom
    L, C, Hh = g.n_patches + 1, g.feat_dim, g.hdm_heads
    P = m.ptab
    dev = tokens.device
    f = "generalFusion."
    T = tokens.dtype
    # the reduced globals replace the cls rows of the key/value streams (AttnMOE.py:155-165)
    for i in range(3):
        ops.set_token0(tokens[i * B * L:(i + 1) * B * L], ori, i * C, 3 * C, B, L)
    sbr_w = P.stride(*[f + b + ".in_proj_weight" for b in BRANCHES])
    sbr_b = P.stride(*[f + b + ".in_proj_bias" for b in BRANCHES])
    sbr_ow = P.stride(*[f + b + ".out_proj.weight" for b in BRANCHES])
    sbr_ob = P.stride(*[f + b + ".out_proj.bias" for b in BRANCHES])
    s_tok = P.stride(*[f + b + "_token" for b in BRANCHES])
    o_w, o_b = P.off(f + "r.in_proj_weight"), P.off(f + "r.in_proj_bias")
    # q[br] = Wq tok + bq and the folded keys U depend on the WEIGHTS only: at inference they are computed once per parameter
    # version (m.weight_cache) instead of on every forward; training recomputes them (the weights move every step)
    cached = m.weight_cache("hdm_qU", T) if tape is None and not training else None
    if cached is not None:
        q, U = cached
    else:
        # q[br] = Wq tok + bq  (batch independent)
        q = torch.empty(7, C, device=dev, dtype=torch.float32)
        ops.gemm_raw(P.flat, P.flat, q, 1, C, C, C, 1, C, 1, C, a_off=P.off(f + "r_token"), b_off=o_w, bias=P.flat, bias_off=o_b,
                     batch1=7, sA=(s_tok, 0), sB=(sbr_w, 0), sC=(C, 0), sBias=(sbr_b, 0))
        # folded keys U[br,h,:] = dh^-0.5 * Wk_h^T q_h   -> rows of a (NC_PAD, C) T matrix, zero padded
        U = m.workspace("hdm_U", (NC_PAD, C), T, zero=True) if tape is not None or training else torch.zeros(NC_PAD, C, device=dev, dtype=T)
        ops.gemm_raw(q, P.flat, U, 1, C, 64, C, 1, 1, C, C, b_off=o_w + C * C, alpha=64 ** -0.5, batch1=7, batch2=Hh,
                     sA=(C, 64), sB=(sbr_w, 64 * C), sC=(Hh * C, C))
        if tape is None and not training:
            m.weight_cache_put("hdm_qU", T, (q, U))
    S = ops.gemm(tokens, U, out_dtype=torch.float32)                              # (3B*L, NC_PAD)
    p_drop = g.hdm_dropout if training else 0.0
    seed = m.next_seed() if p_drop > 0 else 0
    pooled, psum, probs = ops.hdm_softmax_pool(S, tokens, B, L, C, Hh, NC_PAD, p_drop, seed, save_probs=tape is not None,
                                               seed_dev=m.seed_dev if p_drop > 0 else None)
    # per-head value projection o[b,br,h*64+e] = Wv_h pooled[b,br,h,:] (+ bv * psum)
    o = torch.empty(B, 7, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(pooled, P.flat, o, B, 64, C, 7 * Hh * C, 1, C, 1, 7 * C, b_off=o_w + 2 * C * C,
                 bias=P.flat if p_drop == 0 else None, bias_off=o_b + 2 * C, batch1=7, batch2=Hh,
                 sA=(Hh * C, C), sB=(sbr_w, 64 * C), sC=(C, 64), sBias=(sbr_b, 64))
    if p_drop > 0:
        ops.hdm_vbias(o, P.flat, o_b + 2 * C, sbr_b, psum, B, C, Hh)
    feats = torch.empty(B, 7, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(o, P.flat, feats, B, C, C, 7 * C, 1, C, 1, 7 * C, b_off=P.off(f + "r.out_proj.weight"), bias=P.flat,
                 bias_off=P.off(f + "r.out_proj.bias"), batch1=7, sA=(C, 0), sB=(sbr_ow, 0), sC=(C, 0), sBias=(sbr_ob, 0))
    if tape is not None:
        tape.d.update(hdm_q=q, hdm_U=U, hdm_probs=probs, hdm_pooled=pooled, hdm_psum=psum, hdm_o=o, hdm_seed=seed,
                      hdm_seed_dev=m.seed_dev if p_drop > 0 else None, hdm_pdrop=p_drop, tokens=tokens)
    return feats


def cls_globals(m, tokens: Tensor, B: int) -> Tensor:
    """(B, 3C) fp32 = [RGB_global | NI_global | TI_global], the trunk's cls rows (meta_arch.py:91-103)."""
    g = m.geom
    L, C = g.n_patches + 1, g.feat_dim
    out = torch.empty(B, 3 * C, device=tokens.device, dtype=torch.float32)
    for i in range(3):
        ops.get_token0(tokens[i * B * L:(i + 1) * B * L], out, i * C, 3 * C, B, L)
    return out


def dgaf_forward(m, tokens: Tensor, B: int, tape: Optional[Tape] = None) -> Tensor:
    """DGAF-V3 (dual_gated_fusion.py:661-745) on the patch tokens of the three modalities -> (B, 3C) fp32.
    Attention pooling in the folded single-query form (csrc/dgaf.hip): q = Wq query + bq is batch independent, so the K/V
    projections of the 3*B*128 tokens collapse into one skinny score GEMM against 3*heads folded keys and a per-head value
    projection of the POOLED token."""
    g = m.geom
    L, C, nh = g.n_patches + 1, g.feat_dim, g.dgaf_heads
    dh = C // nh
    P = m.ptab
    dev = tokens.device
    T = tokens.dtype
    d = "dgaf."
    sv = tape is not None
    s_q = P.stride(d + "rgb_query", d + "nir_query", d + "tir_query")
    o_w, o_b = P.off(d + "attn_pool.in_proj_weight"), P.off(d + "attn_pool.in_proj_bias")
    cached = m.weight_cache("dgaf_qU", T) if (not sv and not m.training) else None
    if cached is not None:
        q3, U = cached                                  # weights-only quantities, computed once per parameter version at inference
    else:
        # q3[s] = Wq query_s + bq
        q3 = torch.empty(3, C, device=dev, dtype=torch.float32)
        ops.gemm_raw(P.flat, P.flat, q3, 3, C, C, s_q, 1, C, 1, C, a_off=P.off(d + "rgb_query"), b_off=o_w, bias=P.flat, bias_off=o_b)
        # folded keys U[s*nh + h, :] = dh^-0.5 * Wk_h^T q3[s, h]  (zero padded to the MFMA kernel's N granule)
        U = m.workspace("dgaf_U", (NC_PAD, C), T, zero=True) if (sv or m.training) else torch.zeros(NC_PAD, C, device=dev, dtype=T)
        ops.gemm_raw(q3, P.flat, U, 1, C, dh, C, 1, 1, C, C, b_off=o_w + C * C, alpha=dh ** -0.5, batch1=3, batch2=nh, sA=(C, dh),
                     sB=(0, dh * C), sC=(nh * C, C))
        if not sv and not m.training:
            m.weight_cache_put("dgaf_qU", T, (q3, U))
    S = ops.gemm(tokens, U, out_dtype=torch.float32)                              # (3B*L, NC_PAD)
    pooled, probs = ops.dgaf_attn_pool(S, tokens, B, L, C, nh, NC_PAD, save_probs=sv)
    # per-head value projection o[(b,s), h*dh + e] = Wv_h pooled[b,s,h,:] + bv
    o = torch.empty(3 * B, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(pooled, P.flat, o, 3 * B, dh, C, nh * C, 1, C, 1, C, b_off=o_w + 2 * C * C, bias=P.flat, bias_off=o_b + 2 * C,
                 batch1=nh, sA=(C, 0), sB=(dh * C, 0), sC=(dh, 0), sBias=(dh, 0))
    ao = ops.gemm(o, P.f(d + "attn_pool.out_proj.weight"), bias=P.f(d + "attn_pool.out_proj.bias"))
    r = ops.layernorm_fwd(ao, P.f(d + "attn_norm.weight"), P.f(d + "attn_norm.bias"), 1e-5, torch.float32, save=sv)
    h = r[0] if sv else r                                                          # (3B, C) == (B, 3, C)
    E = ops.gemm(h, P.f(d + "entropy_proj.weight"), bias=P.f(d + "entropy_proj.bias"))
    g1 = ops.gemm(h.view(B, 3 * C), P.f(d + "gate_net.0.weight"), bias=P.f(d + "gate_net.0.bias"))
    r1 = ops.layernorm_fwd(g1, P.f(d + "gate_net.1.weight"), P.f(d + "gate_net.1.bias"), 1e-5, torch.float32, save=sv)
    g1n = r1[0] if sv else r1
    alpha = P.f(d + "_alpha").view(1)
    gf = ops.dgaf_gate_fuse(h, E, g1n, P.f(d + "gate_net.3.weight"), P.f(d + "gate_net.3.bias"), alpha, g.dgaf_tau, save=sv)
    fused = gf[0] if sv else gf
    m1 = ops.gemm(fused, P.f(d + "modal_enhance.0.weight"), bias=P.f(d + "modal_enhance.0.bias"))
    r2 = ops.layernorm_fwd(m1, P.f(d + "modal_enhance.1.weight"), P.f(d + "modal_enhance.1.bias"), 1e-5, torch.float32, save=sv)
    enh = r2[0] if sv else r2
    out = ops.dgaf_combine(h, enh)
    if sv:
        tape.d.update(tokens=tokens, dg_q3=q3, dg_U=U, dg_probs=probs, dg_pooled=pooled, dg_o=o, dg_ao=ao, dg_ln=r[1:], dg_h=h,
                      dg_g1=g1, dg_ln1=r1[1:], dg_g1n=g1n, dg_relu=gf[1], dg_aux=gf[2], dg_fused=fused, dg_m1=m1, dg_ln2=r2[1:])
    return out


def atmoe_forward(m, feats: Tensor, B: int, training: bool, tape: Optional[Tape] = None) -> Tensor:
    """ATMoE (AttnMOE.py:94-121): attention gate over the 7 features + 7*H per-chunk experts -> (B, 7C) fp32."""
    g = m.geom
    C, H = g.feat_dim, g.moe_head
    d = C // H
    P, S = m.ptab, m.stab
    dev = feats.device
    gk = "generalFusion.moe.gating_network.gate."
    if not training and tape is None and m.fused_atmoe_eval and ops.atmoe_eval_supported(C, H):
        # inference: the whole module in two launches (csrc/atmoe.hip); BatchNorm = affine map of the running statistics
        e0 = "generalFusion.moe.experts.0.expertHead.0.mlp."
        e1 = "generalFusion.moe.experts.0.expertHead.1.mlp."
        es = P.off(e1 + "0.weight") - P.off(e0 + "0.weight")
        ss = S.off(e1 + "2.running_mean") - S.off(e0 + "2.running_mean")
        ws = m._ws.setdefault(("atmoe_eval", B), {})          # the three workspaces of this batch size
        gfold = m.weight_cache("atmoe_gfold", torch.float32)  # Wq[h]^T Wk[h]: weights only, once per parameter version
        if gfold is None:
            gfold = ops.atmoe_fold_keys(P.f(gk + "q_.weight"), P.f(gk + "k_.weight"), H)
            m.weight_cache_put("atmoe_gfold", torch.float32, gfold)
        bn1 = (P.f(gk + "linear_re.2.weight"), P.f(gk + "linear_re.2.bias"), S.f(gk + "linear_re.2.running_mean"),
               S.f(gk + "linear_re.2.running_var"))
        return ops.atmoe_eval_fwd(feats, P.f(gk + "linear_re.0.weight"), P.f(gk + "linear_re.0.bias"), bn1, gfold,
                                  P.f(e0 + "0.weight"), P.f(e0 + "0.bias"), P.f(e0 + "2.weight"), P.f(e0 + "2.bias"),
                                  S.f(e0 + "2.running_mean"), S.f(e0 + "2.running_var"), es, ss, g.bn_eps, B, C, H, ws=ws)
    # (B, 7C) x (7C, C): a skinny product with a long reduction -> runs split-K (linear epilogue), activation applied after
    x1_pre = ops.gemm(feats.view(B, 7 * C), P.f(gk + "linear_re.0.weight"), bias=P.f(gk + "linear_re.0.bias"))
    x1 = ops.act_fwd(x1_pre, ACT_QUICKGELU)
    bn1 = ops.batchnorm_fwd(x1, P.f(gk + "linear_re.2.weight"), P.f(gk + "linear_re.2.bias"), S.f(gk + "linear_re.2.running_mean"),
                            S.f(gk + "linear_re.2.running_var"), training, g.bn_momentum, g.bn_eps, save=tape is not None)
    x2 = bn1[0] if tape is not None else bn1
    qg = ops.gemm(x2, P.f(gk + "q_.weight"))
    kg = ops.gemm(feats.view(B * 7, C), P.f(gk + "k_.weight"))
    gates = ops.atmoe_gate_softmax(qg, kg, B, C, H)
    e0 = "generalFusion.moe.experts.0.expertHead.0.mlp."
    e1 = "generalFusion.moe.experts.0.expertHead.1.mlp."
    es_w = P.off(e1 + "0.weight") - P.off(e0 + "0.weight")
    y = torch.empty(B, 7 * C, device=dev, dtype=torch.float32)
    ypre = torch.empty(B, 7 * C, device=dev, dtype=torch.float32) if tape is not None else None
    ops.gemm_raw(feats, P.flat, y, B, d, d, 7 * C, 1, d, 1, 7 * C, b_off=P.off(e0 + "0.weight"), bias=P.flat,
                 bias_off=P.off(e0 + "0.bias"), act=ACT_QUICKGELU, batch1=H, batch2=7, sA=(d, C), sB=(7 * es_w, es_w),
                 sC=(d, C), sBias=(7 * es_w, es_w), pre_out=ypre)
    ss = S.off(e1 + "2.running_mean") - S.off(e0 + "2.running_mean")
    bn2 = ops.batchnorm_fwd(y, P.f(e0 + "2.weight"), P.f(e0 + "2.bias"), S.f(e0 + "2.running_mean"), S.f(e0 + "2.running_var"),
                            training, g.bn_momentum, g.bn_eps, save=tape is not None, group=(d, H, 7, es_w, ss), gate=gates)
    moe = bn2[0] if tape is not None else bn2
    if tape is not None:
        tape.d.update(at_x1=x1, at_x1pre=x1_pre, at_bn1=bn1[1:], at_x2=x2, at_q=qg, at_k=kg, at_gates=gates, at_y=y, at_ypre=ypre,
                      at_bn2=bn2[1:], feats=feats)
    return moe


def head_forward(m, x: Tensor, bn: str, cls: str, training: bool, tape: Optional[Tape] = None, tag: str = "") -> Tensor:
    """classifier(bottleneck(x)) (make_model.py:134-138,174-194) -> logits (B, classes) fp32."""
    P, S = m.ptab, m.stab
    g = m.geom
    r = ops.batchnorm_fwd(x, P.f(bn + ".weight"), P.f(bn + ".bias"), S.f(bn + ".running_mean"), S.f(bn + ".running_var"), training,
                          g.bn_momentum, g.bn_eps, save=tape is not None)
    xb = r[0] if tape is not None else r
    score = ops.gemm(xb, P.f(cls + ".weight"))
    if tape is not None:
        tape.d["head_" + tag] = dict(x=x, xb=xb, stats=r[1:], bn=bn, cls=cls)
    return score
