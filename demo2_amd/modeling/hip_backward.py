"""Backward kernel schedule of the hot path and the training-step driver.

The forward in training mode (`train_forward`) records every activation on a Tape (HBM is plentiful: nothing is
recomputed) and returns the reference's output tuple through ONE autograd boundary node; when the loss calls
.backward(), `model_backward` replays the schedule in reverse through libdemo_hip.so.  Parameter gradients are
ACCUMULATED straight into the model's flat fp32 gradient buffer (every p.grad is a view of it), so a data-parallel
run reduces a few large contiguous slices with RCCL while the remaining backward kernels are still running.

Gradient formulas follow the forward definitions cited in hip_forward.py; every step is checked against torch
autograd of the CPU oracle in tests/test_gpu_train.py.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import ops
from ..ops import ACT_GELU, ACT_QUICKGELU
from . import hip_forward as HF
from .hip_forward import BRANCHES, NC_PAD


# ------------------------------------------------------------------------------------------------ autograd boundary
class _Boundary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, tape, anchor, *outs):
        ctx.model, ctx.tape = model, tape
        return tuple(o.detach() for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        model_backward(ctx.model, ctx.tape, list(gouts))
        ctx.tape = None
        return (None, None, None) + tuple(None for _ in gouts)


def train_forward(m, imgs: List[torch.Tensor], cam: Optional[torch.Tensor]):
    """Training-mode forward: the reference's return tuple (engine/processor.py:80-97 pairing), differentiable."""
    from ..dist import ensure_grad_sync
    ensure_grad_sync(m)                    # multi-rank job without an explicit GradSync: install one (see demo2_amd/dist.py)
    m.prepare_grads()
    tape = HF.Tape()
    B = imgs[0].shape[0]
    tape.d["B"] = B
    T = m.compute_dtype
    tokens = HF.trunk_forward(m, imgs, cam, T, (), tape, drop_path=True)
    if m.USE_DGAF:
        feat = HF.dgaf_forward(m, tokens, B, tape)
        outs = m._train_outputs_dgaf(feat, None if m.direct else HF.cls_globals(m, tokens, B), tape)
    else:
        ori = HF.pife_forward(m, tokens, B, tape)
        moe = None
        if m.HDM:
            feats = HF.hdm_forward(m, tokens, ori, B, True, tape)
            moe = HF.atmoe_forward(m, feats, B, True, tape) if m.ATM else feats.view(B, -1)
        outs = m._train_outputs(ori, moe, tape)
    tens = [o for o in outs if torch.is_tensor(o)]
    tape.d["n_out"] = len(tens)
    wrapped = list(_Boundary.apply(m, tape, m.anchor(), *tens))
    return tuple(wrapped.pop(0) if torch.is_tensor(o) else o for o in outs)


# ------------------------------------------------------------------------------------------------ helpers
def _ones(m, n):
    key = ("ones", (n,), torch.float32)
    if key not in m._ws or m._ws[key].device != m.ptab.flat.device:
        m._ws[key] = torch.ones(n, device=m.ptab.flat.device, dtype=torch.float32)
    return m._ws[key]


def _bias_grad(m, src, src_off, N, Bn, src_ld, G, g_off, batch=(1, 1), sSrc=(0, 0), sG=(0, 0)):
    """G[g_off + z*sG + n] += sum_b src[src_off + z*sSrc + b*src_ld + n]  (column sums as a 1 x Bn times Bn x N product,
    so that strided / grouped bias blocks of the flat gradient buffer are reached without scatter kernels)."""
    ones = _ones(m, Bn)
    ops.gemm_raw(ones, src, G, 1, N, Bn, Bn, 1, 1, src_ld, N, b_off=src_off, c_off=g_off, batch1=batch[0], batch2=batch[1],
                 sA=(0, 0), sB=sSrc, sC=sG, accumulate=True)


def _head_backward(m, tape, tag, dscore, dx):
    """classifier(bottleneck(x)) backward; the feature gradient is accumulated into dx (B, F)."""
    P, G = m.ptab, m.grad_buffer()
    h = tape.d["head_" + tag]
    x, xb, (sm, si), bn, cls = h["x"], h["xb"], h["stats"], h["bn"], h["cls"]
    W = P.f(cls + ".weight")                                      # (classes, F)
    ops.gemm_tn(dscore, xb, G, out_off=P.off(cls + ".weight"), ldc=W.shape[1], accumulate=True)
    dxb = ops.gemm(dscore, W.t())                                 # (B, F)
    if m._frozen is None:
        m._frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    db = None if (bn + ".bias") in m._frozen else G[P.off(bn + ".bias"):]     # BNNeck bias is frozen (make_model.py:137,177)
    ops.batchnorm_bwd(dxb, x, P.f(bn + ".weight"), None, sm, si, G[P.off(bn + ".weight"):], db, dx=dx, accumulate_dx=True)


def model_backward(m, tape, gouts):
    _model_backward(m, tape, gouts)
    if m.grad_sync:
        m.grad_sync.backward_done()


def _zeros_like_or(gr, ref):
    return torch.zeros_like(ref) if gr is None else gr.contiguous().float().clone()


def _model_backward_dgaf(m, tape, gouts):
    """Backward of the DGAF-only branch: heads -> DGAF-V3 -> token gradients (+ cls gradients of the per-modality heads)."""
    g = m.geom
    B, C, L = tape.d["B"], g.feat_dim, g.n_patches + 1
    gi = iter(gouts)
    ds, dfe = next(gi), next(gi)
    d_feat = _zeros_like_or(dfe, tape.d["head_dgaf"]["x"])
    if ds is not None:
        _head_backward(m, tape, "dgaf", ds.contiguous().float(), d_feat)
    d_glob = None
    if not m.direct:
        d_glob = torch.zeros(B, 3 * C, device=d_feat.device, dtype=torch.float32)
        for i, s in enumerate("rnt"):
            ds, dfe = next(gi), next(gi)
            dgi = _zeros_like_or(dfe, tape.d["head_" + s]["x"])
            if ds is not None:
                _head_backward(m, tape, s, ds.contiguous().float(), dgi)
            d_glob[:, i * C:(i + 1) * C] = dgi
    dtok = dgaf_backward(m, tape, d_feat, B)
    dtokens = torch.empty(3 * B * L, C, device=d_feat.device, dtype=m.compute_dtype)
    for i in range(3):
        sl = slice(i * B * L, (i + 1) * B * L)
        ops.tokens_grad(dtok[sl], d_glob, i * C, 3 * C, None, 0, 0, False, dtokens[sl], B, L)
    trunk_backward(m, tape, dtokens)


def dgaf_backward(m, tape, dout, B):
    """-> dtok (3B*L, C) fp32: gradient w.r.t. the trunk's tokens through DGAF-V3 (cls rows zero).  Reverse of
    hip_forward.dgaf_forward; parameter gradients are accumulated into the flat gradient buffer."""
    g, P, G, t = m.geom, m.ptab, m.grad_buffer(), tape.d
    L, C, nh = g.n_patches + 1, g.feat_dim, g.dgaf_heads
    dh = C // nh
    dev = dout.device
    d = "dgaf."
    tokens = t["tokens"]
    T = tokens.dtype

    def off(n):
        return P.off(d + n)

    def lin_bwd(dy, x, name, dx_out=None, accumulate=False):
        """y = x W^T + b: dW += dy^T x, db += colsum(dy), returns dx = dy W (optionally accumulated into dx_out)."""
        W = P.f(d + name + ".weight")
        ops.gemm_tn(dy, x, G, out_off=off(name + ".weight"), ldc=W.shape[1], accumulate=True)
        ops.colsum(dy, out=G[off(name + ".bias"):off(name + ".bias") + W.shape[0]], accumulate=True)
        return ops.gemm(dy, W.t(), out=dx_out, accumulate=accumulate)

    dout = dout.contiguous()
    h, g1n, aux = t["dg_h"], t["dg_g1n"], t["dg_aux"]
    alpha = P.f(d + "_alpha").view(1)
    # out = h_s + enh ; enh = LN(modal_enhance.0(fused))
    denh = ops.dgaf_sum3(dout, C)
    mean2, rstd2 = t["dg_ln2"]
    dm1 = ops.layernorm_bwd(denh, t["dg_m1"], P.f(d + "modal_enhance.1.weight"), mean2, rstd2, dw=G[off("modal_enhance.1.weight"):],
                            db=G[off("modal_enhance.1.bias"):])
    dfused = lin_bwd(dm1, t["dg_fused"], "modal_enhance.0")
    # gates + blend
    dhh, dE, dg1n, dlogit, dalpha = ops.dgaf_gate_fuse_bwd(dfused, dout, h, g1n, P.f(d + "gate_net.3.weight"), aux, alpha, g.dgaf_tau)
    ops.gemm_tn(dlogit, t["dg_relu"], G, out_off=off("gate_net.3.weight"), ldc=C, accumulate=True)
    ops.colsum(dlogit, out=G[off("gate_net.3.bias"):off("gate_net.3.bias") + 3], accumulate=True)
    ops.colsum(dalpha.view(B, 1), out=G[off("_alpha"):off("_alpha") + 1], accumulate=True)
    mean1, rstd1 = t["dg_ln1"]
    dg1 = ops.layernorm_bwd(dg1n, t["dg_g1"], P.f(d + "gate_net.1.weight"), mean1, rstd1, dw=G[off("gate_net.1.weight"):],
                            db=G[off("gate_net.1.bias"):])
    lin_bwd(dg1, h.view(B, 3 * C), "gate_net.0", dx_out=dhh.view(B, 3 * C), accumulate=True)
    lin_bwd(dE.view(3 * B, C), h, "entropy_proj", dx_out=dhh.view(3 * B, C), accumulate=True)
    # attention pooling: LayerNorm <- out_proj <- per-head value projection <- pooled tokens
    mean0, rstd0 = t["dg_ln"]
    dao = ops.layernorm_bwd(dhh.view(3 * B, C), t["dg_ao"], P.f(d + "attn_norm.weight"), mean0, rstd0, dw=G[off("attn_norm.weight"):],
                            db=G[off("attn_norm.bias"):])
    do = lin_bwd(dao, t["dg_o"], "attn_pool.out_proj")                                          # (3B, C)
    o_w, o_b = off("attn_pool.in_proj_weight"), off("attn_pool.in_proj_bias")
    pooled = t["dg_pooled"]
    # dWv[h*dh + m, n] += sum_(b,s) do[(b,s), h*dh + m] pooled[b,s,h,n]
    ops.gemm_raw(do, pooled, G, dh, C, 3 * B, 1, C, 1, nh * C, C, c_off=o_w + 2 * C * C, batch1=nh, sA=(dh, 0), sB=(C, 0), sC=(dh * C, 0),
                 accumulate=True)
    ops.colsum(do, out=G[o_b + 2 * C:o_b + 3 * C], accumulate=True)
    dpooled = torch.empty(B, 3, nh, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(do, P.flat, dpooled, 3 * B, C, dh, C, 1, 1, C, nh * C, b_off=o_w + 2 * C * C, batch1=nh, sA=(dh, 0), sB=(dh * C, 0), sC=(C, 0))
    dS, dtok = ops.dgaf_attn_pool_bwd(dpooled, pooled, t["dg_probs"], tokens, B, L, C, nh, NC_PAD)
    # scores S = tokens U^T
    U = t["dg_U"]
    if T == torch.bfloat16:
        dS_lp = ops.cast_f32_bf16(dS.view(-1), torch.empty(dS.numel(), device=dev, dtype=torch.bfloat16)).view_as(dS)
        Ut = ops.transpose(U)
    else:
        dS_lp, Ut = dS, U.t()
    dU = torch.empty(NC_PAD, C, device=dev, dtype=torch.float32)
    ops.gemm_tn(dS_lp, tokens, dU, accumulate=False)
    ops.gemm(dS_lp, Ut, out=dtok, accumulate=True)
    # U[s*nh + h, :] = dh^-0.5 Wk_h^T q3[s, h] ; q3 = Wq query + bq
    q3 = t["dg_q3"]
    scale = dh ** -0.5
    dq3 = torch.empty(3, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(dU, P.flat, dq3, 1, dh, C, C, 1, C, 1, C, b_off=o_w + C * C, alpha=scale, batch1=3, batch2=nh, sA=(nh * C, C), sB=(0, dh * C),
                 sC=(C, dh))
    # dWk[h*dh + m, n] += scale * sum_s q3[s, h*dh + m] dU[s*nh + h, n]   (the three modalities share Wk: reduced inside one product)
    ops.gemm_raw(q3, dU, G, dh, C, 3, 1, C, 1, nh * C, C, c_off=o_w + C * C, alpha=scale, batch1=nh, sA=(dh, 0), sB=(C, 0), sC=(dh * C, 0),
                 accumulate=True)
    s_q = P.stride(d + "rgb_query", d + "nir_query", d + "tir_query")
    o_q = off("rgb_query")
    # dWq[m, n] += sum_s dq3[s, m] query_s[n] ; dbq += colsum(dq3) ; dquery_s = dq3[s] Wq
    ops.gemm_raw(dq3, P.flat, G, C, C, 3, 1, C, 1, s_q, C, b_off=o_q, c_off=o_w, accumulate=True)
    ops.colsum(dq3, out=G[o_b:o_b + C], accumulate=True)
    ops.gemm_raw(dq3, P.flat, G, 3, C, C, C, 1, 1, C, s_q, b_off=o_w, c_off=o_q, accumulate=True)
    return dtok


def _model_backward(m, tape, gouts):
    if m.USE_DGAF:
        return _model_backward_dgaf(m, tape, gouts)
    g = m.geom
    B, C = tape.d["B"], g.feat_dim
    gi = iter(gouts)
    d_moe = None
    has_moe = m.HDM
    # ---- heads -----------------------------------------------------------------------------------------------
    if has_moe:
        ds, dfe = next(gi), next(gi)
        d_moe = _zeros_like_or(dfe, tape.d["head_moe"]["x"])
        if ds is not None:
            _head_backward(m, tape, "moe", ds.contiguous().float(), d_moe)
    if m.direct:
        ds, dfe = next(gi), next(gi)
        d_ori = _zeros_like_or(dfe, tape.d["head_ori"]["x"])
        if ds is not None:
            _head_backward(m, tape, "ori", ds.contiguous().float(), d_ori)
    else:
        d_ori = torch.zeros(B, 3 * C, device=tape.d["head_r"]["x"].device, dtype=torch.float32)
        for i, s in enumerate("rnt"):
            ds, dfe = next(gi), next(gi)
            dgi = _zeros_like_or(dfe, tape.d["head_" + s]["x"])
            if ds is not None:
                _head_backward(m, tape, s, ds.contiguous().float(), dgi)
            d_ori[:, i * C:(i + 1) * C] = dgi
    dtok = None
    if has_moe:
        dfeats = atmoe_backward(m, tape, d_moe, B) if m.ATM else d_moe.view(B, 7, C)
        dtok = hdm_backward(m, tape, dfeats, d_ori, B)
    dtokens = pife_backward(m, tape, d_ori, dtok, B)
    trunk_backward(m, tape, dtokens)


# ------------------------------------------------------------------------------------------------ ATMoE
def atmoe_backward(m, tape, d_moe, B):
    g, P, G, t = m.geom, m.ptab, m.grad_buffer(), tape.d
    C, H = g.feat_dim, g.moe_head
    d = C // H
    dev = d_moe.device
    feats = t["feats"]
    gk = "generalFusion.moe.gating_network.gate."
    e0 = "generalFusion.moe.experts.0.expertHead.0.mlp."
    e1 = "generalFusion.moe.experts.0.expertHead.1.mlp."
    es = P.off(e1 + "0.weight") - P.off(e0 + "0.weight")
    # BN(+gate) of the experts
    dgate = torch.zeros(B, H, 7, device=dev, dtype=torch.float32)
    sm2, si2 = t["at_bn2"]
    dy = ops.batchnorm_bwd(d_moe, t["at_y"], P.f(e0 + "2.weight"), P.f(e0 + "2.bias"), sm2, si2, G[P.off(e0 + "2.weight"):],
                           G[P.off(e0 + "2.bias"):], group=(d, H, 7, es), gate=t["at_gates"], dgate=dgate)
    dypre = ops.act_bwd(dy, t["at_ypre"], ACT_QUICKGELU)
    # expert Linear(d,d): dW[h,j] += dypre_chunk^T feats_chunk ; db += colsum ; dfeats_chunk = dypre_chunk W
    ops.gemm_raw(dypre, feats, G, d, d, B, 1, 7 * C, 1, 7 * C, d, c_off=P.off(e0 + "0.weight"), batch1=H, batch2=7, sA=(d, C), sB=(d, C),
                 sC=(7 * es, es), accumulate=True)
    _bias_grad(m, dypre, 0, d, B, 7 * C, G, P.off(e0 + "0.bias"), batch=(H, 7), sSrc=(d, C), sG=(7 * es, es))
    dfeats = torch.empty(B, 7, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(dypre, P.flat, dfeats, B, d, d, 7 * C, 1, 1, d, 7 * C, b_off=P.off(e0 + "0.weight"), batch1=H, batch2=7, sA=(d, C),
                 sB=(7 * es, es), sC=(d, C))
    # gate: softmax -> q,k projections -> BN -> QuickGELU -> Linear(7C, C)
    dq, dk = ops.atmoe_gate_softmax_bwd(dgate, t["at_gates"], t["at_q"], t["at_k"], B, C, H)
    ops.gemm_tn(dq, t["at_x2"], G, out_off=P.off(gk + "q_.weight"), ldc=C, accumulate=True)
    dx2 = ops.gemm(dq, P.f(gk + "q_.weight").t())
    ops.gemm_tn(dk, feats.view(B * 7, C), G, out_off=P.off(gk + "k_.weight"), ldc=C, accumulate=True)
    ops.gemm(dk, P.f(gk + "k_.weight").t(), out=dfeats.view(B * 7, C), accumulate=True)
    sm1, si1 = t["at_bn1"]
    dx1 = ops.batchnorm_bwd(dx2, t["at_x1"], P.f(gk + "linear_re.2.weight"), None, sm1, si1, G[P.off(gk + "linear_re.2.weight"):],
                            G[P.off(gk + "linear_re.2.bias"):])
    dx1pre = ops.act_bwd(dx1, t["at_x1pre"], ACT_QUICKGELU)
    ops.gemm_tn(dx1pre, feats.view(B, 7 * C), G, out_off=P.off(gk + "linear_re.0.weight"), ldc=7 * C, accumulate=True)
    ops.colsum(dx1pre, out=G[P.off(gk + "linear_re.0.bias"):P.off(gk + "linear_re.0.bias") + C], accumulate=True)
    ops.gemm(dx1pre, P.f(gk + "linear_re.0.weight").t(), out=dfeats.view(B, 7 * C), accumulate=True)
    return dfeats


# ------------------------------------------------------------------------------------------------ HDM
def hdm_backward(m, tape, dfeats, d_ori, B):
    """-> dtok (3B*L, C) fp32: gradient w.r.t. the HDM token streams (rows 0 already routed into d_ori)."""
    g, P, G, t = m.geom, m.ptab, m.grad_buffer(), tape.d
    L, C, Hh = g.n_patches + 1, g.feat_dim, g.hdm_heads
    dev = dfeats.device
    f = "generalFusion."
    tokens = t["tokens"]
    T = tokens.dtype
    sw = P.stride(*[f + b + ".in_proj_weight" for b in BRANCHES])
    sb = P.stride(*[f + b + ".in_proj_bias" for b in BRANCHES])
    sow = P.stride(*[f + b + ".out_proj.weight" for b in BRANCHES])
    sob = P.stride(*[f + b + ".out_proj.bias" for b in BRANCHES])
    stok = P.stride(*[f + b + "_token" for b in BRANCHES])
    o_w, o_b = P.off(f + "r.in_proj_weight"), P.off(f + "r.in_proj_bias")
    o_ow, o_ob = P.off(f + "r.out_proj.weight"), P.off(f + "r.out_proj.bias")
    dfeats = dfeats.contiguous()
    o = t["hdm_o"]
    # out_proj
    ops.gemm_raw(dfeats, o, G, C, C, B, 1, 7 * C, 1, 7 * C, C, c_off=o_ow, batch1=7, sA=(C, 0), sB=(C, 0), sC=(sow, 0), accumulate=True)
    _bias_grad(m, dfeats, 0, C, B, 7 * C, G, o_ob, batch=(7, 1), sSrc=(C, 0), sG=(sob, 0))
    do = torch.empty(B, 7, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(dfeats, P.flat, do, B, C, C, 7 * C, 1, 1, C, 7 * C, b_off=o_ow, batch1=7, sA=(C, 0), sB=(sow, 0), sC=(C, 0))
    # per-head value projection
    pooled, psum, pdrop = t["hdm_pooled"], t["hdm_psum"], t["hdm_pdrop"]
    ops.gemm_raw(do, pooled, G, 64, C, B, 1, 7 * C, 1, 7 * Hh * C, C, c_off=o_w + 2 * C * C, batch1=7, batch2=Hh, sA=(C, 64), sB=(Hh * C, C),
                 sC=(sw, 64 * C), accumulate=True)
    dpsum = None
    if pdrop > 0:
        dpsum = torch.zeros(B, 7, Hh, device=dev, dtype=torch.float32)
        ops.hdm_vbias_bwd(do, P.flat, o_b + 2 * C, sb, psum, G, dpsum, B, C, Hh)
    else:
        _bias_grad(m, do, 0, C, B, 7 * C, G, o_b + 2 * C, batch=(7, 1), sSrc=(C, 0), sG=(sb, 0))
    dpooled = torch.empty(B, 7, Hh, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(do, P.flat, dpooled, B, C, 64, 7 * C, 1, 1, C, 7 * Hh * C, b_off=o_w + 2 * C * C, batch1=7, batch2=Hh, sA=(C, 64),
                 sB=(sw, 64 * C), sC=(Hh * C, C))
    # softmax + pooling
    dS, dtok = ops.hdm_softmax_pool_bwd(dpooled, dpsum, t["hdm_probs"], tokens, pooled, psum, B, L, C, Hh, NC_PAD, pdrop, t["hdm_seed"],
                                        seed_dev=t.get("hdm_seed_dev"))
    # scores S = tokens U^T
    U = t["hdm_U"]
    if T == torch.bfloat16:
        dS_lp = ops.cast_f32_bf16(dS.view(-1), torch.empty(dS.numel(), device=dev, dtype=torch.bfloat16)).view_as(dS)
        Ut = ops.transpose(U)
    else:
        dS_lp, Ut = dS, U.t()
    dU = torch.empty(NC_PAD, C, device=dev, dtype=torch.float32)
    ops.gemm_tn(dS_lp, tokens, dU, accumulate=False)
    ops.gemm(dS_lp, Ut, out=dtok, accumulate=True)
    # U[(br,h), :] = dh^-0.5 * Wk_h^T q_h ;  q = Wq tok + bq
    q = t["hdm_q"]
    scale = 64 ** -0.5
    dq = torch.empty(7, C, device=dev, dtype=torch.float32)
    ops.gemm_raw(dU, P.flat, dq, 1, 64, C, C, 1, C, 1, C, b_off=o_w + C * C, alpha=scale, batch1=7, batch2=Hh, sA=(Hh * C, C), sB=(sw, 64 * C),
                 sC=(C, 64))
    ops.gemm_raw(q, dU, G, 64, C, 1, 1, 1, 1, 1, C, c_off=o_w + C * C, alpha=scale, batch1=7, batch2=Hh, sA=(C, 64), sB=(Hh * C, C),
                 sC=(sw, 64 * C), accumulate=True)
    ops.gemm_raw(dq, P.flat, G, C, C, 1, 1, 1, 1, 1, C, b_off=P.off(f + "r_token"), c_off=o_w, batch1=7, sA=(C, 0), sB=(stok, 0), sC=(sw, 0),
                 accumulate=True)
    _bias_grad(m, dq, 0, C, 1, C, G, o_b, batch=(7, 1), sSrc=(C, 0), sG=(sb, 0))
    ops.gemm_raw(dq, P.flat, G, 1, C, C, C, 1, 1, C, C, b_off=o_w, c_off=P.off(f + "r_token"), batch1=7, sA=(C, 0), sB=(sw, 0), sC=(stok, 0),
                 accumulate=True)
    # row 0 of every stream is the (PIFE) global feature: route its gradient into d_ori
    for i in range(3):
        ops.token0_grad(dtok[i * B * L:(i + 1) * B * L], d_ori, i * C, 3 * C, B, L)
    return dtok


# ------------------------------------------------------------------------------------------------ PIFE
def pife_backward(m, tape, d_ori, dtok, B):
    """-> dtokens (3B*L, C) in the compute dtype: gradient w.r.t. the trunk output."""
    g, P, G, t = m.geom, m.ptab, m.grad_buffer(), tape.d
    L, C = g.n_patches + 1, g.feat_dim
    dev = d_ori.device
    T = m.compute_dtype
    dtokens = torch.empty(3 * B * L, C, device=dev, dtype=T)
    if not g.global_local:
        for i in range(3):
            sl = slice(i * B * L, (i + 1) * B * L)
            ops.tokens_grad(None if dtok is None else dtok[sl], d_ori, i * C, 3 * C, None, 0, 0, False, dtokens[sl], B, L)
        return dtokens
    dpre = ops.act_bwd(d_ori, t["pife_pre"], ACT_QUICKGELU)                       # (B, 3C)
    zl, z = t["pife_zl"], t["pife_z"]
    s_w = P.stride("rgb_reduce.1.weight", "nir_reduce.1.weight", "tir_reduce.1.weight")
    s_b = P.stride("rgb_reduce.1.bias", "nir_reduce.1.bias", "tir_reduce.1.bias")
    o_w, o_b = P.off("rgb_reduce.1.weight"), P.off("rgb_reduce.1.bias")
    ops.gemm_raw(dpre, zl, G, C, 2 * C, B, 1, 3 * C, 1, 2 * C, 2 * C, c_off=o_w, batch1=3, sA=(C, 0), sB=(B * 2 * C, 0), sC=(s_w, 0),
                 accumulate=True)
    _bias_grad(m, dpre, 0, C, B, 3 * C, G, o_b, batch=(3, 1), sSrc=(C, 0), sG=(s_b, 0))
    dzl = torch.empty(3 * B, 2 * C, device=dev, dtype=torch.float32)
    ops.gemm_raw(dpre, P.flat, dzl, B, 2 * C, C, 3 * C, 1, 1, 2 * C, 2 * C, b_off=o_w, batch1=3, sA=(C, 0), sB=(s_w, 0), sC=(B * 2 * C, 0))
    dz = torch.empty(3 * B, 2 * C, device=dev, dtype=torch.float32)
    for i, name in enumerate(("rgb_reduce", "nir_reduce", "tir_reduce")):
        mean, rstd = t["pife_stats"][i]
        sl = slice(i * B, (i + 1) * B)
        ops.layernorm_bwd(dzl[sl], z[sl], P.f(name + ".0.weight"), mean, rstd, dx=dz[sl], dw=G[P.off(name + ".0.weight"):],
                          db=G[P.off(name + ".0.bias"):])
    for i in range(3):
        sl = slice(i * B * L, (i + 1) * B * L)
        dzi = dz[i * B:(i + 1) * B]
        ops.tokens_grad(None if dtok is None else dtok[sl], dzi, 0, 2 * C, dzi, C, 2 * C, False, dtokens[sl], B, L)
    return dtokens


# ------------------------------------------------------------------------------------------------ trunk
def trunk_backward(m, tape, dtokens):
    g, P, G, t = m.geom, m.ptab, m.grad_buffer(), tape.d
    B = t["B"]
    Bn, np_, L, W = 3 * B, g.n_patches, g.n_patches + 1, g.width
    T = m.compute_dtype
    dev = dtokens.device
    lp = T == torch.bfloat16
    act = ACT_QUICKGELU if g.arch == "clip" else ACT_GELU
    sync = m.grad_sync if m.grad_sync else None        # (False = explicit opt-out, see dist.ensure_grad_sync)

    def gw(name):
        return G[P.off(name):]

    # ---- final projection + LayerNorm --------------------------------------------------------------------------
    if g.arch == "clip":
        y = t["y"]
        ops.gemm_tn(y, dtokens, G, out_off=P.off(g.k_proj), ldc=g.feat_dim, accumulate=True)      # dproj (W, C) = y^T dtokens
        dy = ops.gemm(dtokens, P.w(g.k_proj, T))                                                   # (3BL, W) = dtokens proj^T
    else:
        dy = dtokens
    # bf16 gradient stream (CLIP trunk, bf16 mode; DEMO_BF16_GRAD_STREAM=0 turns it off): the residual gradient lives only as bf16
    # between the LayerNorm backwards (demo_layernorm_bwd_lp: 380 instead of 608 MB per launch) -- one rounding per LayerNorm instead
    # of a fp32 stream rounded once for the GEMMs.  Measured at full size against the fp32 mode (tools/probes/grad_stream_accuracy.py):
    # per-parameter gradient error median 2.72 % / p90 4.51 % / max 23.5 %, whole gradient 2.68 % -- the same figures to three digits
    # as with the fp32 stream (2.71 / 4.51 / 23.5 / 2.68): the bf16 operands of the GEMMs set the error, not the stream; 41.2 -> 40.4 ms.
    lps = bool(lp and m.bf16_grad_stream and g.k_lnpre and not any(b_.get("dp_a") is not None for b_ in tape.blocks if b_))
    dx = None if lps else torch.empty(Bn * L, W, device=dev, dtype=torch.float32)
    dxb = torch.empty(Bn * L, W, device=dev, dtype=torch.bfloat16) if lp else None
    if lps:
        ops.layernorm_bwd_lp(dy, t["x_final"], P.f(g.k_lnpost + ".weight"), t["meanf"], t["rstdf"], None, dxb, dw=gw(g.k_lnpost + ".weight"),
                             db=gw(g.k_lnpost + ".bias"))
    else:
        ops.layernorm_bwd(dy, t["x_final"], P.f(g.k_lnpost + ".weight"), t["meanf"], t["rstdf"], dx=dx, dw=gw(g.k_lnpost + ".weight"),
                          db=gw(g.k_lnpost + ".bias"), dx_lp=dxb)
    if sync is not None:
        sync.tail_ready()
    # ---- blocks in reverse ---------------------------------------------------------------------------------------------
    # Weight gradients (dW = dY^T X) feed nothing in the backward chain, so in bf16 mode they go to a SIDE stream: their
    # workgroups fill the CUs the dX GEMMs leave idle in their last partial round (582 tiles on 256 CUs = 2.27 rounds for the
    # N = 768 products) and the dispatch gaps between the dependent launches of the main chain.  Opt-in (DEMO_OVERLAP_DW=1):
    # measured 41.9 vs 42.5 ms per step at B = 128 -- both GEMM families hold a whole CU's LDS per workgroup, so the two
    # queues mostly alternate instead of sharing CUs -- and per-kernel timings (bench.py roofline, rocprofv3) stop being
    # the kernels' own rates once they overlap, so the default stays single-stream.
    side = m.side_stream() if (lp and m.overlap_dw and not torch.cuda.is_current_stream_capturing()) else None
    main = torch.cuda.current_stream() if side is not None else None
    # Grouped weight gradients (round 3, bf16 gradient stream without the side stream): the four dW products of a block are issued as
    # ONE launch at the end of the block's backward (ops.gemm_tn_group / demo_gemm_tn_bf16_group) -- their tails, partial-tile store
    # bursts and launch ramps overlap instead of adding up.  Their dY operands must then outlive two LayerNorm backwards, so the
    # bf16 gradient stream rotates through THREE buffers instead of being updated in place.  DEMO_GROUP_DW=0 restores one launch per Linear.
    group_dw = bool(lps and side is None and m.group_dw)
    dxbs = [dxb, torch.empty_like(dxb)] if side is not None else ([dxb, torch.empty_like(dxb), torch.empty_like(dxb)] if group_dw else [dxb, dxb])
    readers = [None, None]                                                      # last side-stream reader of each of them
    cur = 0
    pending = []                                                                # this block's deferred (dY, X, weight key, ldc, bias key)

    def flush_dw():
        if pending:
            order = reversed(pending) if m.group_dw_fresh_first else pending
            ops.gemm_tn_group([(a_t, b_t, P.off(kw), ldc, P.off(kb)) for a_t, b_t, kw, ldc, kb in order], G, accumulate=True)
            pending.clear()

    def dw(a_t, b_t, key_w, ldc, key_b, buf=None):
        """dW (+ bias gradient) of one Linear; on the side stream when there is one."""
        if group_dw:
            pending.append((a_t, b_t, key_w, ldc, key_b))
            return
        if side is None:
            ops.gemm_tn(a_t, b_t, G, out_off=P.off(key_w), ldc=ldc, accumulate=True, colsum_out=G, colsum_off=P.off(key_b))
            return
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.gemm_tn(a_t, b_t, G, out_off=P.off(key_w), ldc=ldc, accumulate=True, colsum_out=G, colsum_off=P.off(key_b))
            if buf is not None:
                readers[buf] = torch.cuda.Event()
                readers[buf].record(side)
        a_t.record_stream(side)                    # the caching allocator must not hand these out again before the side
        b_t.record_stream(side)                    # stream has read them

    def next_dxb():
        """bf16 gradient buffer the next LayerNorm backward may overwrite: its last reader on the side stream has finished."""
        nonlocal cur
        if group_dw:                            # three buffers in rotation: the two a block's deferred dW products still read stay intact
            cur = (cur + 1) % 3
            return dxbs[cur]
        cur ^= 1
        if side is not None and readers[cur] is not None:
            main.wait_event(readers[cur])
            readers[cur] = None
        return dxbs[cur]

    for i in reversed(range(g.layers)):
        k, s = g.block_keys(i), tape.blocks[i]
        d_res = dxbs[cur] if lp else dx                             # gradient of the block output, GEMM operand dtype
        rbuf = cur
        if s.get("dp_m") is not None:                               # DropPath: the branch sees scale[image] * gradient
            d_res, rbuf = ops.rowscale_cast(dx, s["dp_m"], Bn, L, T), None
        # x2 = x1 + drop_path(fc2(u))
        dw(d_res, s["u"], k["fc2_w"], 4 * W, k["fc2_b"], buf=rbuf)
        dpre = ops.gemm(d_res, P.wt(k["fc2_w"], T), act_grad_pre=s["pre"],                          # (M, 4W) = (dx W2) * act'(pre)
                        act_grad_kind=ops.ACT_SAVED if s.get("pre_deriv") else act)
        # u = act(fc1(h2))
        dw(dpre, s["h2"], k["fc1_w"], W, k["fc1_b"])
        dh2 = ops.gemm(dpre, P.wt(k["fc1_w"], T))                                                  # (M, W)
        prev = dxbs[cur] if lp else None
        nb = next_dxb() if lp else None
        if lps:
            ops.layernorm_bwd_lp(dh2, s["x1"], P.f(k["ln2"] + ".weight"), s["mean2"], s["rstd2"], prev, nb, dw=gw(k["ln2"] + ".weight"),
                                 db=gw(k["ln2"] + ".bias"))
        else:
            ops.layernorm_bwd(dh2, s["x1"], P.f(k["ln2"] + ".weight"), s["mean2"], s["rstd2"], dx=dx, accumulate_dx=True,
                              dw=gw(k["ln2"] + ".weight"), db=gw(k["ln2"] + ".bias"), dx_lp=nb)
        d_res = nb if lp else dx
        rbuf = cur
        if s.get("dp_a") is not None:
            d_res, rbuf = ops.rowscale_cast(dx, s["dp_a"], Bn, L, T), None
        # x1 = x + drop_path(out_proj(a))
        dw(d_res, s["a"], k["proj_w"], W, k["proj_b"], buf=rbuf)
        da = ops.gemm(d_res, P.wt(k["proj_w"], T))                                                 # (M, W)
        dqkv = ops.attention_bwd(s["qkv"], s["a"], da, s["lse"], Bn, L, g.heads)                   # (M, 3W)
        dw(dqkv, s["h1"], k["qkv_w"], W, k["qkv_b"])
        dh1 = ops.gemm(dqkv, P.wt(k["qkv_w"], T))                                                  # (M, W)
        prev = dxbs[cur] if lp else None
        nb = next_dxb() if lp else None
        if lps:
            ops.layernorm_bwd_lp(dh1, s["x"], P.f(k["ln1"] + ".weight"), s["mean1"], s["rstd1"], prev, nb, dw=gw(k["ln1"] + ".weight"),
                                 db=gw(k["ln1"] + ".bias"))
        else:
            ops.layernorm_bwd(dh1, s["x"], P.f(k["ln1"] + ".weight"), s["mean1"], s["rstd1"], dx=dx, accumulate_dx=True,
                              dw=gw(k["ln1"] + ".weight"), db=gw(k["ln1"] + ".bias"), dx_lp=nb)
        flush_dw()                                                  # the block's four weight gradients, one launch
        tape.blocks[i] = None                                       # release the activations of this block
        if sync is not None:
            if side is not None:
                main.wait_stream(side)                              # this block's weight gradients are final
            sync.block_ready(i)
    if side is not None:
        main.wait_stream(side)
    # ---- embedding ----------------------------------------------------------------------------------------------------
    if g.k_lnpre:
        dpre = ops.layernorm_bwd(dxbs[cur] if lps else dx, t["emb_pre"], P.f(g.k_lnpre + ".weight"), t["emb_mean"], t["emb_rstd"], dw=gw(g.k_lnpre + ".weight"),
                                 db=gw(g.k_lnpre + ".bias"))
    else:
        dpre = dx
    dpatch = ops.embed_tokens_bwd(dpre, t["cam3"], g.sie_coe, g.cam_all_tokens, T, gw(g.k_cls), gw(g.k_pos),
                                  gw(g.cam_embed) if g.cam_embed else None, Bn, np_,
                                  n_cam=P.index[g.cam_embed][1][0] if g.cam_embed else 0, err=m.device_errors() if g.cam_embed else None)
    ops.gemm_tn(dpatch, t["cols"], G, out_off=P.off(g.k_conv), ldc=3 * g.patch * g.patch, accumulate=True)
    if g.k_conv_bias:
        ops.colsum(dpatch, out=G[P.off(g.k_conv_bias):P.off(g.k_conv_bias) + W], accumulate=True)


# ------------------------------------------------------------------------------------------------ training step driver
def label_smooth_xent(logits, target, eps=0.1):
    """CrossEntropyLabelSmooth (layers/softmax_loss.py:23-34); eps = 0 is F.cross_entropy."""
    logp = torch.log_softmax(logits.float(), 1)
    K = logits.shape[1]
    t = torch.full_like(logp, eps / K).scatter_(1, target[:, None], 1 - eps + eps / K)
    return (-t * logp).mean(0).sum()


def triplet_soft_margin(feat, labels):
    """TripletLoss() without margin: batch-hard mining + SoftMarginLoss (layers/triplet_loss.py:51-135)."""
    feat = feat.float()
    sq = (feat ** 2).sum(1, keepdim=True)
    dist = (sq + sq.t() - 2 * feat @ feat.t()).clamp(min=1e-12).sqrt()
    same = labels[:, None] == labels[None, :]
    d_ap = torch.where(same, dist, dist.new_full((), float("-inf"))).max(1).values
    d_an = torch.where(~same, dist, dist.new_full((), float("inf"))).min(1).values
    return torch.nn.functional.softplus(-(d_an - d_ap)).mean()


def loss_settings(cfg):
    """What layers/make_loss.py:13-80 builds from the configuration -> dict(idw, trw, eps, margin):
      DATALOADER.SAMPLER 'softmax'         : loss = F.cross_entropy(score, target) only (:32-34; no weights, no smoothing, no triplet)
      DATALOADER.SAMPLER 'softmax_triplet' : ID_LOSS_WEIGHT * xent + TRIPLET_LOSS_WEIGHT * triplet (:37-73) with
          xent    = CrossEntropyLabelSmooth(eps 0.1) if MODEL.IF_LABELSMOOTH == 'on' else F.cross_entropy  (eps 0)
          triplet = TripletLoss() soft margin if MODEL.NO_MARGIN else TripletLoss(SOLVER.MARGIN)          (:18-23)
    margin None = soft margin.  Everything else the reference only prints a message for and then fails at the first step."""
    sampler = cfg.DATALOADER.SAMPLER
    if sampler == "softmax":
        return dict(idw=1.0, trw=0.0, eps=0.0, margin=None)
    if sampler != "softmax_triplet":
        raise NotImplementedError(f"DATALOADER.SAMPLER={sampler!r}: make_loss builds no loss function for it (layers/make_loss.py:75-78)")
    if cfg.MODEL.METRIC_LOSS_TYPE != "triplet":
        raise NotImplementedError(f"MODEL.METRIC_LOSS_TYPE={cfg.MODEL.METRIC_LOSS_TYPE!r}: only 'triplet' has a loss in make_loss (:38,:71-73)")
    return dict(idw=float(cfg.MODEL.ID_LOSS_WEIGHT), trw=float(cfg.MODEL.TRIPLET_LOSS_WEIGHT),
                eps=0.1 if cfg.MODEL.IF_LABELSMOOTH == "on" else 0.0,
                margin=None if cfg.MODEL.NO_MARGIN else float(cfg.SOLVER.MARGIN))


def triplet_margin(feat, labels, margin):
    """TripletLoss(margin): batch-hard mining + MarginRankingLoss (layers/triplet_loss.py:114-117,131-134)."""
    feat = feat.float()
    sq = (feat ** 2).sum(1, keepdim=True)
    dist = (sq + sq.t() - 2 * feat @ feat.t()).clamp(min=1e-12).sqrt()
    same = labels[:, None] == labels[None, :]
    d_ap = torch.where(same, dist, dist.new_full((), float("-inf"))).max(1).values
    d_an = torch.where(~same, dist, dist.new_full((), float("inf"))).min(1).values
    return torch.clamp(d_ap - d_an + margin, min=0).mean()


def make_loss_fn(cfg):
    """layers/make_loss.py:13-80 as a torch function (the checker of the fused loss kernels, and TrainStep(fused_loss=False))."""
    st = loss_settings(cfg)

    def loss_fn(score, feat, target, target_cam=None):
        loss = st["idw"] * label_smooth_xent(score, target, st["eps"])
        if st["trw"] != 0.0:
            loss = loss + st["trw"] * (triplet_soft_margin(feat, target) if st["margin"] is None else triplet_margin(feat, target, st["margin"]))
        return loss
    return loss_fn


def total_loss(outputs, target, loss_fn):
    """Pairing rule of engine/processor.py:80-97."""
    n = len(outputs) - (len(outputs) % 2)
    loss = 0
    for i in range(0, n, 2):
        loss = loss + loss_fn(outputs[i], outputs[i + 1], target)
    if len(outputs) % 2 == 1:
        loss = loss + outputs[-1]
    return loss


def fused_loss_and_grads(outputs, target, cfg, err=None):
    """The same total (pairing rule of engine/processor.py:80-97, make_loss 'softmax_triplet' with label smoothing and the
    soft-margin triplet) through the loss kernels (csrc/loss.hip): returns (loss (1,) fp32, [d loss / d output_i] for the tensor
    outputs), ready for torch.autograd.backward(outputs, grads) -- the loss itself never enters autograd."""
    st = loss_settings(cfg)
    n = len(outputs) - (len(outputs) % 2)
    loss = torch.zeros(1, device=outputs[0].device, dtype=torch.float32)
    roots, grads = [], []
    for i in range(0, n, 2):
        score, feat = outputs[i].detach(), outputs[i + 1].detach()
        grads.append(ops.xent_label_smooth(score.float().contiguous(), target, loss, st["eps"], st["idw"], err=err))
        roots.append(outputs[i])
        if st["trw"] != 0.0:
            grads.append(ops.triplet_soft_margin(feat.float().contiguous(), target, loss, st["trw"],
                                                 margin=-1.0 if st["margin"] is None else st["margin"]))
            roots.append(outputs[i + 1])
    if len(outputs) % 2 == 1 and torch.is_tensor(outputs[-1]):
        loss = loss + outputs[-1].detach().float().reshape(1)
        if outputs[-1].requires_grad:
            roots.append(outputs[-1])
            grads.append(torch.ones_like(outputs[-1]))
    return loss, roots, grads


class TrainStep:
    """One optimisation step of the reference's do_train loop (engine/processor.py:57-104) on resident tensors:
    forward, loss (layers/make_loss.py as configured), backward (with in-backward RCCL gradient reduction when world > 1), fused flat
    optimizer step with the reference's parameter groups (solver/make_optimizer.py: SOLVER.OPTIMIZER_NAME Adam / AdamW / SGD,
    WEIGHT_DECAY_BIAS, LARGE_FC_LR; frozen and never-used parameters are left alone as torch leaves them).  Configuration values
    without an implementation raise NotImplementedError here instead of training something else silently.

    With graph=True (single-GPU runs) the whole step -- ~650 kernel launches, half of them tiny (fusion, heads, loss) -- is
    captured ONCE into a hipGraph and replayed: the host then issues one graph launch per step instead of driving every
    launch from Python, which removes the launch gaps the GPU otherwise idles in (~3 ms of a 47 ms step).  Everything that
    changes between steps lives in device memory: the batch (static input buffers), Adam's learning rate and bias
    corrections (3 floats refreshed by an async copy before each replay) and the HDM dropout seed (a device counter that a
    captured kernel advances)."""

    def __init__(self, model, cfg, world=1, lr=None, optimizer=True, graph=False, fused_loss=True):
        from ..dist import GradSync
        self.model, self.cfg = model, cfg
        self.fused_loss = bool(fused_loss)     # loss value + output gradients from csrc/loss.hip instead of the torch restatement
        self.loss_fn = make_loss_fn(cfg)
        self.lr = float(cfg.SOLVER.BASE_LR if lr is None else lr)
        self.wd = float(cfg.SOLVER.WEIGHT_DECAY)
        self.optimizer = optimizer
        self.t = 0
        loss_settings(cfg)                                   # raises for loss configurations make_loss has no function for
        if "center" in str(cfg.MODEL.METRIC_LOSS_TYPE) or str(getattr(cfg.MODEL, "IF_WITH_CENTER", "no")) == "yes":
            raise NotImplementedError("center loss (engine/processor.py:106-110) is outside the PIFE -> HDM -> ATMoE path")
        name = str(cfg.SOLVER.OPTIMIZER_NAME)
        kinds = {"Adam": ops.OPT_ADAM, "AdamW": ops.OPT_ADAMW, "SGD": ops.OPT_SGD}
        if optimizer and name not in kinds:
            raise NotImplementedError(f"SOLVER.OPTIMIZER_NAME={name!r}: the flat step implements Adam, AdamW and SGD (solver/make_optimizer.py:19-24)")
        self.opt_kind = kinds.get(name, ops.OPT_ADAM)
        self.momentum = float(cfg.SOLVER.MOMENTUM)
        self.groups = None                                   # built on the model's device at the first step
        model.grad_sync = GradSync(model) if world > 1 else None
        flat = model.ptab.flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        if graph and self.opt_kind == ops.OPT_SGD:
            raise NotImplementedError("graph-captured SGD steps are not built (no device-side hyper-parameters for SGD)")
        self.use_graph = bool(graph) and world == 1 and flat.is_cuda
        self._graph = None
        self._static = None
        self._eager_steps = 0
        if self.use_graph:
            self.hyper = torch.zeros(3, device=flat.device, dtype=torch.float32)       # {lr, update count t, unused}
            self._lr_uploaded = None
            model.seed_dev = torch.zeros(1, device=flat.device, dtype=torch.int64)

    # ---- one step, as a sequence of stream operations (eager, or under capture) ----------------------------------------
    def _run(self, x, target, cam, hyper_dev):
        model = self.model
        out = model(x, label=target, cam_label=cam)
        if self.fused_loss:
            loss, roots, grads = fused_loss_and_grads(out, target, self.cfg, err=model.device_errors())
            torch.autograd.backward(roots, grads)
            loss = loss[0]
        else:
            loss = total_loss(out, target, self.loss_fn)
            loss.backward()
        if self.optimizer:
            if hyper_dev is not None:
                hyper_dev[1:2].add_(1.0)                               # the device-side update count t
            if self.groups is None or self.groups[0].device != model.ptab.flat.device:
                s = self.cfg.SOLVER
                self.groups = model.param_groups(self.wd, float(s.WEIGHT_DECAY_BIAS), bool(s.LARGE_FC_LR))
            b1 = self.momentum if self.opt_kind == ops.OPT_SGD else 0.9
            model.optim_step(self.opt_kind, self.m, self.v, self.lr, b1, 0.999, 1e-8, self.wd, max(self.t, 1), hyper_dev=hyper_dev,
                             groups=self.groups)
        if model.seed_dev is not None:
            model.seed_dev.add_(0x9E3779B97F4A7C15 - (1 << 64))       # next dropout stream (int64 wrap-around)
        model.device_errors().post()                                  # out-of-range labels / camera ids of this step (polled at the next forward)
        return loss

    def _refresh_hyper(self):
        """Learning rate -> device memory when it changed (a stream-ordered fill); the update count lives on the device."""
        if self._lr_uploaded != self.lr:
            self.hyper[0:1].fill_(self.lr)
            self._lr_uploaded = self.lr

    def step(self, x, target, cam):
        if self.optimizer:
            self.t += 1
        if not self.use_graph:
            return self._run(x, target, cam, None)
        self._refresh_hyper()
        if self._graph is None:
            # a few eager steps first (workspaces, transposed shadows and gradient views get allocated outside the graph's
            # private pool), then one capture
            if self._eager_steps < 2:
                self._eager_steps += 1
                return self._run(x, target, cam, self.hyper)
            self._static = ({k: v.clone() for k, v in x.items()}, target.clone(), cam.clone() if cam is not None else None)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._loss = self._run(self._static[0], self._static[1], self._static[2], self.hyper)
        sx, st, sc = self._static
        for k, v in x.items():
            if v.data_ptr() != sx[k].data_ptr():
                sx[k].copy_(v, non_blocking=True)
        st.copy_(target, non_blocking=True)
        if sc is not None:
            sc.copy_(cam, non_blocking=True)
        self._graph.replay()
        return self._loss


def train_step_smoke(m, xd, cam):
    """One fwd+bwd+Adam step on the smoke batch; checks that the loss is finite and the parameters moved."""
    from ..config import get_defaults
    m.train()
    B = cam.shape[0]
    target = torch.arange(B, device=cam.device) // 2
    ts = TrainStep(m, m.cfg, 1)
    before = m.ptab.flat.clone()
    loss = ts.step(xd, target, cam)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item(), loss
    moved = (m.ptab.flat - before).abs().max().item()
    print(f"smoke: train step loss = {loss.item():.4f}, max |dparam| = {moved:.2e}")
    assert moved > 0
    m.eval()
