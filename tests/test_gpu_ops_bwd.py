"""GPU: backward C-ABI entry points against torch autograd of the same op on CPU (fp32)."""
import pytest
import torch
import torch.nn.functional as F

from demo2_amd import ops
from demo2_amd._lib import ACT_GELU, ACT_QUICKGELU
from oracle import demo_oracle as O
from test_gpu_ops import STREAMS, attn_ref, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("R,Mo,No", [(24768 // 16, 256, 128), (129 * 3, 384, 128), (1000, 128, 512), (64, 128, 128), (37, 128, 128)])
def test_gemm_tn_bf16_mfma(R, Mo, No):
    a, b = rnd(R, Mo).bfloat16(), rnd(R, No, seed=1).bfloat16()
    want = a.float().t() @ b.float()
    base = rnd(Mo, No, seed=2)
    out = base.clone().to(DEV)
    ops.gemm_tn(a.to(DEV), b.to(DEV), out, accumulate=True)
    torch.testing.assert_close(out.cpu(), base + want, rtol=2e-3, atol=2e-3 * R ** 0.5)
    out2 = torch.full((Mo, No), 7.0, device=DEV)
    cs = torch.ones(Mo + 8, device=DEV)                       # fused bias gradient: colsum of the A operand
    ops.gemm_tn(a.to(DEV), b.to(DEV), out2, accumulate=False, alpha=0.5, colsum_out=cs, colsum_off=4)
    torch.testing.assert_close(out2.cpu(), 0.5 * want, rtol=2e-3, atol=2e-3 * R ** 0.5)
    torch.testing.assert_close(cs.cpu()[4:4 + Mo], 1 + a.float().sum(0), rtol=1e-3, atol=1e-3 * R ** 0.5)
    assert torch.equal(cs.cpu()[:4], torch.ones(4)) and torch.equal(cs.cpu()[4 + Mo:], torch.ones(4))
    # exact integer check of the transposing fragment reads (asymmetric operands)
    a2 = ((torch.arange(R)[:, None] * 3 + torch.arange(Mo)[None, :]) % 5).float()
    b2 = ((torch.arange(R)[:, None] + 2 * torch.arange(No)[None, :]) % 3).float()
    out3 = torch.zeros(Mo, No, device=DEV)
    ops.gemm_tn(a2.bfloat16().to(DEV), b2.bfloat16().to(DEV), out3, accumulate=False)
    assert torch.equal(out3.cpu(), a2.t() @ b2)


def test_gemm_tn_group_equals_single_launches():
    """demo_gemm_tn_bf16_group: the weight gradients of a block (four Linears of different shapes over the same reduction rows, with
    bias gradients, written at offsets of one flat buffer) as ONE launch equal the four single launches to fp32 summation order (same
    kernel; the group has its own split of the reduction), and both equal the fp32 product; an item outside the group contract falls back."""
    R, W = 4100, 256                                         # ragged reduction length: the last stage of every slice is partial
    shapes = [(W, 4 * W), (4 * W, W), (3 * W, W), (W, W)]    # (Mo, No): fc2, fc1, qkv, proj in miniature
    g = torch.Generator().manual_seed(0)
    ops_ab = [((torch.randn(R, mo, generator=g) * 0.5).bfloat16().to(DEV), (torch.randn(R, no, generator=g) * 0.5).bfloat16().to(DEV)) for mo, no in shapes]
    sizes = [mo * no for mo, no in shapes]
    w_off = [sum(sizes[:i]) + 64 * i for i in range(4)]
    b_off = [w_off[3] + sizes[3] + 64 + 2048 * i for i in range(4)]
    n_flat = b_off[3] + 4096
    base = torch.randn(n_flat, generator=g)
    flat_g, flat_s = base.clone().to(DEV), base.clone().to(DEV)
    ops.gemm_tn_group([(a, b, w_off[i], shapes[i][1], b_off[i]) for i, (a, b) in enumerate(ops_ab)], flat_g, accumulate=True)
    for i, (a, b) in enumerate(ops_ab):
        ops.gemm_tn(a, b, flat_s, out_off=w_off[i], ldc=shapes[i][1], accumulate=True, colsum_out=flat_s, colsum_off=b_off[i])
    # same kernel and arithmetic; the group cuts the reduction into its own number of slices (one count for all products), so the fp32 sums
    # differ from the single launches' in their order only
    torch.testing.assert_close(flat_g.cpu(), flat_s.cpu(), rtol=1e-5, atol=2e-3)
    for i, (a, b) in enumerate(ops_ab):
        mo, no = shapes[i]
        want = base[w_off[i]: w_off[i] + mo * no].view(mo, no) + a.float().cpu().t() @ b.float().cpu()
        torch.testing.assert_close(flat_g.cpu()[w_off[i]: w_off[i] + mo * no].view(mo, no), want, rtol=2e-3, atol=2e-3 * R ** 0.5)
        torch.testing.assert_close(flat_g.cpu()[b_off[i]: b_off[i] + mo], base[b_off[i]: b_off[i] + mo] + a.float().cpu().sum(0), rtol=1e-3,
                                   atol=1e-3 * R ** 0.5)
    # untouched gaps of the flat buffer
    assert torch.equal(flat_g.cpu()[w_off[0] + sizes[0]: w_off[1]], base[w_off[0] + sizes[0]: w_off[1]])
    # fallback: an item with Mo = 128 is outside the 256^2 contract
    a5, b5 = (torch.randn(R, 128, generator=g)).bfloat16().to(DEV), (torch.randn(R, 256, generator=g)).bfloat16().to(DEV)
    out5 = torch.zeros(128 * 256, device=DEV)
    ops.gemm_tn_group([(a5, b5, 0, 256, None)], out5, accumulate=False)
    torch.testing.assert_close(out5.cpu().view(128, 256), a5.float().cpu().t() @ b5.float().cpu(), rtol=2e-3, atol=2e-3 * R ** 0.5)


def test_gemm_tn_generic_and_offset():
    a, b = rnd(50, 33), rnd(50, 20, seed=1)
    flat = torch.zeros(4000, device=DEV)
    ops.gemm_tn(a.to(DEV), b.to(DEV), flat, out_off=100, ldc=24, accumulate=True)
    got = flat[100:100 + 33 * 24].reshape(33, 24)[:, :20]
    torch.testing.assert_close(got.cpu(), a.t() @ b, rtol=1e-4, atol=1e-4)
    assert flat[:100].abs().sum() == 0


@pytest.mark.parametrize("Bn,L,H,dt", [(2, 129, 12, torch.bfloat16), (2, 9, 2, torch.bfloat16), (1, 144, 2, torch.bfloat16),
                                       (1, 160, 1, torch.bfloat16), (2, 129, 12, torch.float32), (2, 9, 2, torch.float32),
                                       (1, 144, 2, torch.float32)])          # fp32 (generic kernel) keeps K,V,dK,dV in LDS: L <= 148
def test_attention_bwd(Bn, L, H, dt):
    D = H * 64
    qkv = rnd(Bn * L, 3 * D).to(dt)
    dout = rnd(Bn * L, D, seed=3).to(dt)
    qr = qkv.float().requires_grad_(True)
    o_ref, _ = attn_ref(qr, Bn, L, H)
    o_ref.backward(dout.float())
    out, lse = ops.attention_fwd(qkv.to(DEV), Bn, L, H, save_lse=True)
    dqkv = ops.attention_bwd(qkv.to(DEV), out, dout.to(DEV), lse, Bn, L, H)
    tol = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dqkv.float().cpu(), qr.grad, **tol)


def test_embed_tokens_bwd():
    Bn, np_, W = 5, 8, 128
    dpre = rnd(Bn * (np_ + 1), W)
    cam = torch.tensor([1, 0, 2, 1, 1])
    for all_tok in (False, True):
        dcls, dpos, dcam = torch.ones(W, device=DEV), torch.ones(np_ + 1, W, device=DEV), torch.ones(3, W, device=DEV)
        dpatch = ops.embed_tokens_bwd(dpre.to(DEV), cam.to(DEV), 1.5, all_tok, torch.float32, dcls, dpos, dcam, Bn, np_, n_cam=3)
        d3 = dpre.reshape(Bn, np_ + 1, W)
        torch.testing.assert_close(dpatch.cpu(), d3[:, 1:].reshape(-1, W), rtol=0, atol=0)
        torch.testing.assert_close(dcls.cpu(), 1 + d3[:, 0].sum(0), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dpos.cpu(), 1 + d3.sum(0), rtol=1e-5, atol=1e-5)
        src = d3.sum(1) if all_tok else d3[:, 0]
        want = torch.ones(3, W).index_add_(0, cam, 1.5 * src)
        torch.testing.assert_close(dcam.cpu(), want, rtol=1e-5, atol=1e-5)


def test_act_bwd():
    for kind, fn in ((ACT_QUICKGELU, O.quick_gelu), (ACT_GELU, F.gelu)):
        x = rnd(1000).requires_grad_(True)
        dy = rnd(1000, seed=1)
        fn(x).backward(dy)
        torch.testing.assert_close(ops.act_bwd(dy.to(DEV), x.detach().to(DEV), kind).cpu(), x.grad, rtol=1e-4, atol=1e-5)


def test_batchnorm_bwd_plain_and_grouped():
    Bn, Fd = 16, 96
    x = (rnd(Bn, Fd) * 2 + 1).requires_grad_(True)
    w, b = (1 + rnd(Fd, seed=1, scale=0.1)).requires_grad_(True), rnd(Fd, seed=2, scale=0.1).requires_grad_(True)
    dy = rnd(Bn, Fd, seed=3)
    F.batch_norm(x, None, None, w, b, True, 0.1, 1e-5).backward(dy)
    rm, rv = torch.zeros(Fd, device=DEV), torch.ones(Fd, device=DEV)
    _, sm, si = ops.batchnorm_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), rm, rv, True, 0.1, 1e-5, save=True)
    dw, db = torch.zeros(Fd, device=DEV), torch.zeros(Fd, device=DEV)
    dx = ops.batchnorm_bwd(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), sm, si, dw, db)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-5)
    # grouped + gate (ATMoE expert BatchNorms)
    H, d, J = 2, 8, 7
    Cc = H * d
    pstride = 3 * d + 5
    x = (rnd(Bn, J * Cc) + 0.2).requires_grad_(True)
    pw, pb = rnd(H * J * pstride, seed=5).requires_grad_(True), rnd(H * J * pstride, seed=6).requires_grad_(True)
    gate = torch.rand(Bn, H, J, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    dy = rnd(Bn, J * Cc, seed=8)
    cols = []
    for j in range(J):
        for h in range(H):
            blk = h * J + j
            y = F.batch_norm(x[:, j * Cc + h * d: j * Cc + (h + 1) * d], None, None, pw[blk * pstride:blk * pstride + d],
                             pb[blk * pstride:blk * pstride + d], True, 0.1, 1e-5)
            cols.append(y * gate[:, h, j, None])
    torch.cat(cols, 1).backward(dy)
    srm, srv = torch.zeros(H * J * 16, device=DEV), torch.ones(H * J * 16, device=DEV)
    _, sm, si = ops.batchnorm_fwd(x.detach().to(DEV), pw.detach().to(DEV), pb.detach().to(DEV), srm, srv, True, 0.1, 1e-5, save=True,
                                  group=(d, H, J, pstride, 16), gate=gate.detach().to(DEV))
    dpw, dpb = torch.zeros(H * J * pstride, device=DEV), torch.zeros(H * J * pstride, device=DEV)
    dgate = torch.zeros(Bn, H, J, device=DEV)
    dx = ops.batchnorm_bwd(dy.to(DEV), x.detach().to(DEV), pw.detach().to(DEV), pb.detach().to(DEV), sm, si, dpw, dpb,
                           group=(d, H, J, pstride), gate=gate.detach().to(DEV), dgate=dgate)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dpw.cpu(), pw.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dpb.cpu(), pb.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dgate.cpu(), gate.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("Bs,Cd,H", [(5, 128, 4), (3, 512, 8)])
def test_atmoe_gate_softmax_bwd(Bs, Cd, H):
    d = Cd // H
    q, k = rnd(Bs, Cd).requires_grad_(True), rnd(Bs * 7, Cd, seed=1).requires_grad_(True)
    g = (torch.einsum("bhd,bjhd->bhj", q.reshape(Bs, H, d), k.reshape(Bs, 7, H, d)) * d ** -0.5).softmax(-1)
    dg = rnd(Bs, H, 7, seed=2)
    g.backward(dg)
    dq, dk = ops.atmoe_gate_softmax_bwd(dg.to(DEV), g.detach().to(DEV), q.detach().to(DEV), k.detach().to(DEV), Bs, Cd, H)
    torch.testing.assert_close(dq.cpu(), q.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dk.cpu(), k.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("Bs,L,Cd,Hh,dt,pd", [(3, 9, 128, 2, torch.float32, 0.0), (2, 129, 512, 8, torch.bfloat16, 0.0),
                                              (2, 17, 128, 2, torch.float32, 0.1), (2, 5, 768, 12, torch.float32, 0.1)])
def test_hdm_softmax_pool_bwd(Bs, L, Cd, Hh, dt, pd):
    ldS = 128
    S = rnd(3 * Bs * L, ldS, scale=2.0)
    tok = rnd(3 * Bs * L, Cd, seed=1).to(dt)
    pooled, psum, probs, post = ops.hdm_softmax_pool(S.to(DEV), tok.to(DEV), Bs, L, Cd, Hh, ldS, p_drop=pd, seed=99, save_probs=True,
                                                     save_probs_post=True)
    mask = (post != 0).float().cpu()
    Sr = S.clone().requires_grad_(True)
    tr = tok.float().requires_grad_(True)
    tf, Sf = tr.reshape(3, Bs, L, Cd), Sr.reshape(3, Bs, L, ldS)
    pl, ps = [], []
    for br, st in enumerate(STREAMS):
        sc = torch.cat([Sf[m][:, :, br * Hh:(br + 1) * Hh] for m in st], 1)
        x = torch.cat([tf[m] for m in st], 1)
        p = sc.softmax(1).transpose(1, 2)                                                  # (Bs, Hh, Lk)
        if pd > 0:
            p = p * mask[:, br, :, :len(st) * L] / (1 - pd)
        pl.append(torch.einsum("bhl,blc->bhc", p, x))
        ps.append(p.sum(-1))
    pooled_ref, psum_ref = torch.stack(pl, 1), torch.stack(ps, 1)
    dpooled, dpsum = rnd(Bs, 7, Hh, Cd, seed=4), rnd(Bs, 7, Hh, seed=5)
    ((pooled_ref * dpooled).sum() + (psum_ref * dpsum).sum()).backward()
    dS, dtok = ops.hdm_softmax_pool_bwd(dpooled.to(DEV), dpsum.to(DEV), probs, tok.to(DEV), pooled, psum, Bs, L, Cd, Hh, ldS, p_drop=pd, seed=99)
    tol = dict(rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dS.cpu(), Sr.grad, **tol)
    torch.testing.assert_close(dtok.cpu(), tr.grad, **tol)


def test_hdm_vbias_bwd_and_token_grads():
    Bs, Cd, Hh = 4, 128, 2
    dy, bias, psum = rnd(Bs, 7, Cd), rnd(7 * 200, seed=3), torch.rand(Bs, 7, Hh, generator=torch.Generator().manual_seed(1))
    dbias, dpsum = torch.zeros(7 * 200, device=DEV), torch.zeros(Bs, 7, Hh, device=DEV)
    ops.hdm_vbias_bwd(dy.to(DEV), bias.to(DEV), 10, 200, psum.to(DEV), dbias, dpsum, Bs, Cd, Hh)
    bsel = torch.stack([bias[10 + br * 200: 10 + br * 200 + Cd] for br in range(7)])            # (7, C)
    want_db = (dy * psum.repeat_interleave(Cd // Hh, -1)).sum(0)
    got_db = torch.stack([dbias.cpu()[10 + br * 200: 10 + br * 200 + Cd] for br in range(7)])
    torch.testing.assert_close(got_db, want_db, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dpsum.cpu(), (dy * bsel[None]).reshape(Bs, 7, Hh, Cd // Hh).sum(-1), rtol=1e-5, atol=1e-5)
    Bn, L = 3, 9
    dtok, dori = rnd(Bn * L, Cd), rnd(Bn, 3 * Cd, seed=1)
    o = dori.clone().to(DEV)
    ops.token0_grad(dtok.to(DEV), o, Cd, 3 * Cd, Bn, L)
    want = dori.clone()
    want[:, Cd:2 * Cd] += dtok.reshape(Bn, L, Cd)[:, 0]
    torch.testing.assert_close(o.cpu(), want, rtol=1e-6, atol=1e-6)
    dz = rnd(Bn, 2 * Cd, seed=2)
    for dt in (torch.float32, torch.bfloat16):
        out = torch.empty(Bn * L, Cd, device=DEV, dtype=dt)
        ops.tokens_grad(dtok.to(DEV), dz.to(DEV), 0, 2 * Cd, dz.to(DEV), Cd, 2 * Cd, False, out, Bn, L)
        w3 = dtok.reshape(Bn, L, Cd).clone()
        w3[:, 0] = dz[:, :Cd]
        w3[:, 1:] += dz[:, None, Cd:] / (L - 1)
        torch.testing.assert_close(out.float().cpu().reshape(Bn, L, Cd), w3.to(dt).float(), rtol=1e-6, atol=1e-6)
    out = torch.empty(Bn * L, Cd, device=DEV)
    ops.tokens_grad(dtok.to(DEV), dz.to(DEV), 0, 2 * Cd, None, 0, 0, True, out, Bn, L)
    w3 = dtok.reshape(Bn, L, Cd).clone()
    w3[:, 0] += dz[:, :Cd]
    torch.testing.assert_close(out.cpu().reshape(Bn, L, Cd), w3, rtol=1e-6, atol=1e-6)
