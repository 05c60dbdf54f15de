"""GPU: every C-ABI entry point against a CPU fp32 computation of the same op (torch on CPU / the oracle).
fp32 paths: tolerance = fp32 re-association noise.  bf16 paths: inputs are rounded to bf16 first, products are
accumulated in fp32 by the MFMA, so the tolerance is the bf16 OUTPUT rounding (2^-8 relative)."""
import math

import pytest
import torch
import torch.nn.functional as F

from demo2_amd import ops
from demo2_amd._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU
from oracle import demo_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def act_ref(x, act):
    if act == ACT_QUICKGELU:
        return O.quick_gelu(x)
    if act == ACT_GELU:
        return F.gelu(x)
    return x


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("tile", [0, 32, 64])
@pytest.mark.parametrize("M,N,K", [(37, 171, 100), (1, 512, 64), (129, 64, 17), (300, 256, 192)])
@pytest.mark.parametrize("act", [ACT_NONE, ACT_QUICKGELU, ACT_GELU])
def test_gemm_generic_fp32(M, N, K, act, tile):
    """tile: the generic kernel's output tile (demo_gemm_generic_tile: 32 / 64 forced, 0 = the dispatcher's choice)."""
    from demo2_amd import _lib
    lib = _lib.load()
    a, b, bias, res = rnd(M, K), rnd(N, K, seed=1), rnd(N, seed=2), rnd(M, N, seed=3)
    want = act_ref(0.5 * a @ b.t() + bias, act) + res
    lib.demo_gemm_generic_tile(tile)
    try:
        got = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), act=act, alpha=0.5)
        # a transposed (column-major) B view and a split-K shape (long reduction, few tiles) through the same tile
        got_t = ops.gemm(a.to(DEV), b.t().contiguous().to(DEV).t(), bias=bias.to(DEV), residual=res.to(DEV), act=act, alpha=0.5)
        a2, b2 = rnd(64, 1536, seed=5), rnd(96, 1536, seed=6, scale=1536 ** -0.5)
        got_sk = ops.gemm(a2.to(DEV), b2.to(DEV))
    finally:
        lib.demo_gemm_generic_tile(0)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got_t.cpu(), want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got_sk.cpu(), a2 @ b2.t(), rtol=1e-4, atol=1e-4)


def test_gemm_generic_strides_batches_and_epilogues():
    # transposed operands through strides
    a, b = rnd(50, 70), rnd(70, 33, seed=1)          # want a @ b  -> B[n,k] = b[k,n]
    got = ops.gemm(a.to(DEV), b.to(DEV).t())
    torch.testing.assert_close(got.cpu(), a @ b, rtol=1e-4, atol=1e-4)
    at = rnd(70, 50, seed=2)                          # A given as (K, M)
    got = ops.gemm(at.to(DEV).t(), b.to(DEV).t())
    torch.testing.assert_close(got.cpu(), at.t() @ b, rtol=1e-4, atol=1e-4)
    # two-level batching with bias: C[z1,z2] = A[z1,z2] @ B[z1,z2]^T + bias[z1,z2]
    Z1, Z2, M, N, K = 3, 4, 5, 6, 7
    A, B, bias = rnd(Z1, Z2, M, K), rnd(Z1, Z2, N, K, seed=1), rnd(Z1, Z2, N, seed=2)
    Cd = torch.empty(Z1, Z2, M, N, device=DEV)
    ops.gemm_raw(A.to(DEV), B.to(DEV), Cd, M, N, K, K, 1, K, 1, N, bias=bias.to(DEV), batch1=Z1, batch2=Z2,
                 sA=(Z2 * M * K, M * K), sB=(Z2 * N * K, N * K), sC=(Z2 * M * N, M * N), sBias=(Z2 * N, N))
    torch.testing.assert_close(Cd.cpu(), A @ B.transpose(-1, -2) + bias[:, :, None, :], rtol=1e-4, atol=1e-4)
    # pre_out, act-grad epilogue, accumulate
    a, b, bias = rnd(40, 30), rnd(20, 30, seed=1), rnd(20, seed=2)
    out, pre = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=ACT_QUICKGELU, pre_out_new=True)
    torch.testing.assert_close(pre.cpu(), a @ b.t() + bias, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.cpu(), O.quick_gelu(a @ b.t() + bias), rtol=1e-4, atol=1e-4)
    for kind in (ACT_QUICKGELU, ACT_GELU):
        p = rnd(40, 20, seed=5).requires_grad_(True)
        act_ref(p, kind).sum().backward()
        got = ops.gemm(a.to(DEV), b.to(DEV), act_grad_pre=p.detach().to(DEV), act_grad_kind=kind)
        torch.testing.assert_close(got.cpu(), (a @ b.t()) * p.grad, rtol=1e-4, atol=1e-4)
    acc = rnd(40, 20, seed=9)
    got = ops.gemm(a.to(DEV), b.to(DEV), out=acc.to(DEV), accumulate=True)
    torch.testing.assert_close(got.cpu(), acc + a @ b.t(), rtol=1e-4, atol=1e-4)
    # mixed dtypes: fp32 activations x bf16 weights
    got = ops.gemm(a.to(DEV), b.to(DEV).bfloat16(), out_dtype=torch.float32)
    torch.testing.assert_close(got.cpu(), a @ b.bfloat16().float().t(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (128, 128, 64), (1161, 384, 128), (24768 // 8, 2304, 768)])
def test_gemm_bf16_mfma(M, N, K):
    a, b, bias, res = rnd(M, K).bfloat16(), rnd(N, K, seed=1, scale=K ** -0.5).bfloat16(), rnd(N, seed=2), rnd(M, N, seed=3)
    ref = a.float() @ b.float().t()
    got = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=ACT_QUICKGELU)                   # bf16 out
    assert got.dtype == torch.bfloat16
    torch.testing.assert_close(got.float().cpu(), O.quick_gelu(ref + bias), rtol=1e-2, atol=1e-2)
    got, pre = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.float32, alpha=2.0,
                        pre_out_new=True)
    torch.testing.assert_close(got.cpu(), 2 * ref + bias + res, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(pre.cpu(), 2 * ref + bias, rtol=1e-3, atol=1e-3)
    # asymmetric-operand check of the fragment maps: A = identity block, B asymmetric integers (exact in bf16)
    if K == 128 and M >= 128:
        a2 = torch.zeros(M, K)
        a2[:K, :K] = torch.eye(K)
        b2 = (torch.arange(N)[:, None] * 2 + torch.arange(K)[None, :] % 7).float()
        got = ops.gemm(a2.bfloat16().to(DEV), b2.bfloat16().to(DEV), out_dtype=torch.float32)
        assert torch.equal(got.cpu(), a2.bfloat16().float() @ b2.bfloat16().float().t())


@pytest.mark.parametrize("reg_epi,tile", [(1, 2566), (0, 2566), (1, 2576), (1, 644), (1, 1603)])
@pytest.mark.parametrize("M,N,K", [(700, 512, 192), (1300, 256, 64), (256, 768, 128), (3 * 256 + 5, 384, 64)])
def test_gemm_bf16_256_kernel_epilogues(M, N, K, reg_epi, tile):
    """The 256x256 ping-pong kernel forced (the heuristic only takes it on chip-filling shapes), both epilogue forms: the register
    epilogue stages the weight rows in a permuted order, so an exact asymmetric-integer product checks the column map; then
    every epilogue variant the path uses (bias, QuickGELU + saved pre-activation, fp32 residual, act'(pre), accumulate),
    ragged M, and the row split."""
    from demo2_amd import _lib
    lib = _lib.load()
    if K % 64 or (tile == 2566 and N % 256):
        pytest.skip("shape outside this tile configuration's contract")
    lib.demo_gemm_set_pp_epilogue(reg_epi)
    lib.demo_gemm_force_tile(tile)            # 2566: 256x256 ping-pong kernel; 2576: 256x128 twin kernel (two workgroups per CU, BK = 32); 644: 64x64, 4 stages (small M); 1603: 160x128, 3 stages (row-split remainders)
    try:
        a2 = torch.zeros(M, K)
        a2[:K, :K] = torch.eye(K)
        a2[K:, :] = (torch.arange(M - K)[:, None] % 3 - 1).float() * (torch.arange(K)[None, :] % 2)
        b2 = (torch.arange(N)[:, None] * 2 + torch.arange(K)[None, :] % 7).float() % 251
        got = ops.gemm(a2.bfloat16().to(DEV), b2.bfloat16().to(DEV), out_dtype=torch.float32)
        assert torch.equal(got.cpu(), a2.bfloat16().float() @ b2.bfloat16().float().t())
        a, b, bias, res = rnd(M, K).bfloat16(), rnd(N, K, seed=1, scale=K ** -0.5).bfloat16(), rnd(N, seed=2), rnd(M, N, seed=3)
        ref = a.float() @ b.float().t()
        ad, bd = a.to(DEV), b.to(DEV)
        got, pre = ops.gemm(ad, bd, bias=bias.to(DEV), act=ACT_QUICKGELU, pre_out_new=True)                       # fc1 forward
        assert got.dtype == torch.bfloat16
        torch.testing.assert_close(got.float().cpu(), O.quick_gelu(ref + bias), rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(pre.float().cpu(), ref + bias, rtol=1e-2, atol=1e-2)
        got = ops.gemm(ad, bd, bias=bias.to(DEV), residual=res.to(DEV), out_dtype=torch.float32, alpha=0.5)       # proj / fc2 forward
        torch.testing.assert_close(got.cpu(), 0.5 * ref + bias + res, rtol=1e-3, atol=1e-3)
        r16 = res.bfloat16().to(DEV)                                                                               # the same on a bf16 residual stream,
        got = ops.gemm(ad, bd, bias=bias.to(DEV), residual=r16, out=r16)                                           # updated in place
        torch.testing.assert_close(got.float().cpu(), ref + bias + res.bfloat16().float(), rtol=1e-2, atol=2e-2)
        for kind in (ACT_QUICKGELU, ACT_GELU):                                                                  # MLP backward
            pr = rnd(M, N, seed=5).bfloat16().float().requires_grad_(True)
            act_ref(pr, kind).sum().backward()
            got = ops.gemm(ad, bd, act_grad_pre=pr.detach().bfloat16().to(DEV), act_grad_kind=kind)
            torch.testing.assert_close(got.float().cpu(), ref * pr.grad, rtol=2e-2, atol=2e-2)
        acc = rnd(M, N, seed=9)
        got = ops.gemm(ad, bd, out=acc.to(DEV), accumulate=True)
        torch.testing.assert_close(got.cpu(), acc + ref, rtol=1e-3, atol=1e-3)
    finally:
        lib.demo_gemm_force_tile(0)
        lib.demo_gemm_set_pp_epilogue(1)


@pytest.mark.parametrize("force,dt,stage", [(0, torch.float32, 1), (1282, torch.bfloat16, 1), (2566, torch.bfloat16, 1), (2566, torch.bfloat16, 0),
                                            (2576, torch.bfloat16, 1), (2562, torch.bfloat16, 1), (1603, torch.bfloat16, 1)])
@pytest.mark.parametrize("kind", [ACT_QUICKGELU, ACT_GELU])
def test_gemm_saved_activation_derivative_d8(force, dt, stage, kind):
    """pre_out_kind 2 / dtPre DEMO_D8 (demo_hip.h): the forward epilogue saves act'(x) as one byte per element, q = rint((a + 0.13) * 200);
    the backward product decodes a = q * 0.005 - 0.13 and multiplies.  Checked on the generic kernel, the 128^2 kernel, the 256^2 kernel
    (register epilogue with and without the LDS image of the tile, LDS-staged epilogue, 256 x 128 twin), ragged M included:
    the codes against the exact derivative (|error| <= half a step), the activation output unchanged, the backward product EXACTLY
    equal (to the output's rounding) to the plain product times the decoded tensor, and the code range of both activations (the
    saturated derivatives 0 and 1 are codes 26 and 226)."""
    from demo2_amd import _lib
    from demo2_amd._lib import ACT_SAVED
    lib = _lib.load()
    M, N, K = (300, 256, 192) if force in (0, 1282) else (2 * 256 + 77, 512, 128)
    lp = dt == torch.bfloat16
    a, b, bias = rnd(M, K) * 2.0, rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2)
    a[:4] *= 40.0                                          # saturated rows: act' -> 0 or 1 exactly
    if lp:
        a, b = a.bfloat16(), b.bfloat16()
    x = (a.float() @ b.float().t() + bias).requires_grad_(True)
    y_ref = act_ref(x, kind)
    y_ref.sum().backward()
    lib.demo_gemm_force_tile(force)
    lib.demo_gemm_set_epi_stage(stage)
    lib.demo_gemm_set_pp_epilogue(0 if force == 2562 else 1)
    try:
        fc = 2566 if force == 2562 else force
        lib.demo_gemm_force_tile(fc)
        y, codes = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=kind, pre_out_new=True, pre_out_deriv=True, pre_out_d8=True)
        assert codes.dtype == torch.uint8 and codes.shape == (M, N)
        tol = 1e-2 if lp else 1e-5
        torch.testing.assert_close(y.float().cpu(), y_ref.detach(), rtol=tol, atol=tol)
        dec = codes.cpu().float() * 0.005 - 0.13
        # half a code step, plus what the bf16 operands' rounding moves the pre-activation (fp32 accumulation order only in the fp32 case)
        err = (dec - x.grad).abs().max().item()
        assert err <= 0.0025 + (2e-3 if lp else 1e-5), err
        assert int(codes.max()) <= 252 and int(codes.min()) >= 0
        sat = x.detach().abs() > 30.0
        assert sat.any() and (dec[sat] - (x.detach()[sat] > 0).float()).abs().max().item() < 1e-6       # codes 26 / 226: 0 and 1 to fp32 rounding
        # the backward product on the codes == the product times the decoded tensor
        dy, w2t = rnd(M, 64, seed=4), rnd(N, 64, seed=5, scale=0.125)
        if lp:
            dy, w2t = dy.bfloat16(), w2t.bfloat16()
        g_saved = ops.gemm(dy.to(DEV), w2t.to(DEV), act_grad_pre=codes, act_grad_kind=ACT_SAVED)
        ref = (dy.float() @ w2t.float().t()) * dec
        torch.testing.assert_close(g_saved.float().cpu(), ref, rtol=tol, atol=tol)
        with pytest.raises(TypeError):
            ops.gemm(dy.to(DEV), w2t.to(DEV), act_grad_pre=codes, act_grad_kind=kind)          # codes are a derivative, not a pre-activation
    finally:
        lib.demo_gemm_force_tile(0)
        lib.demo_gemm_set_epi_stage(1)
        lib.demo_gemm_set_pp_epilogue(1)


@pytest.mark.parametrize("force,dt", [(0, torch.float32), (1282, torch.bfloat16), (2566, torch.bfloat16)])
@pytest.mark.parametrize("kind", [ACT_QUICKGELU, ACT_GELU])
def test_gemm_saved_activation_derivative(force, dt, kind):
    """pre_out_kind 1 (demo_hip.h): the forward epilogue saves act'(x) instead of the pre-activation x; the backward product then takes
    act_grad_kind DEMO_ACT_SAVED -- one multiply.  On the generic fp32 kernel and on both MFMA kernels (256^2: forced, incl. a ragged M),
    against torch autograd of the activation."""
    from demo2_amd import _lib
    from demo2_amd._lib import ACT_SAVED
    lib = _lib.load()
    M, N, K = (300, 256, 192) if force != 2566 else (2 * 256 + 77, 512, 128)
    lp = dt == torch.bfloat16
    a, b, bias = rnd(M, K), rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2)
    if lp:
        a, b = a.bfloat16(), b.bfloat16()
    x = (a.float() @ b.float().t() + bias).requires_grad_(True)
    y_ref = act_ref(x, kind)
    y_ref.sum().backward()
    lib.demo_gemm_force_tile(force)
    try:
        y, dsave = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=kind, pre_out_new=True, pre_out_deriv=True)
        tol = 1e-2 if lp else 1e-5
        torch.testing.assert_close(y.float().cpu(), y_ref.detach(), rtol=tol, atol=tol)
        torch.testing.assert_close(dsave.float().cpu(), x.grad, rtol=tol, atol=tol)
        # backward product on the saved derivative == the same product re-evaluating act' from a saved pre-activation
        dy, w2t = rnd(M, 64, seed=4), rnd(N, 64, seed=5, scale=0.125)
        if lp:
            dy, w2t = dy.bfloat16(), w2t.bfloat16()
        g_saved = ops.gemm(dy.to(DEV), w2t.to(DEV), act_grad_pre=dsave, act_grad_kind=ACT_SAVED)
        ref = (dy.float() @ w2t.float().t()) * dsave.float().cpu()
        torch.testing.assert_close(g_saved.float().cpu(), ref, rtol=tol, atol=tol)
        _, pre = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=kind, pre_out_new=True)
        g_eval = ops.gemm(dy.to(DEV), w2t.to(DEV), act_grad_pre=pre, act_grad_kind=kind)
        torch.testing.assert_close(g_saved.float().cpu(), g_eval.float().cpu(), rtol=3e-2 if lp else 1e-5, atol=3e-2 if lp else 1e-5)
    finally:
        lib.demo_gemm_force_tile(0)


def test_gemm_bf16_row_split_matches_single_launch():
    from demo2_amd import _lib
    lib = _lib.load()
    M, N, K = 162 * 256 - 7, 768, 64           # 486 tiles of 256^2 = 1.9 rounds: the heuristic takes the 256^2 kernel, 230 tiles left over
    a, b, bias = rnd(M, K).bfloat16().to(DEV), rnd(N, K, seed=1, scale=K ** -0.5).bfloat16().to(DEV), rnd(N, seed=2).to(DEV)
    outs = []
    try:
        for left in (0, 250):
            lib.demo_gemm_set_split(left)
            lib.demo_gemm_force_tile(0)
            outs.append(ops.gemm(a, b, bias=bias, out_dtype=torch.float32).cpu())
    finally:
        lib.demo_gemm_set_split(128)
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(outs[1], a.float().cpu() @ b.float().cpu().t() + bias.cpu(), rtol=1e-3, atol=1e-3)


def test_gemm_bf16_equal_parts_plan():
    """Round 3: a product whose 256^2 tile count leaves a partial last round is issued as full + 1 EQUAL launches of the 256^2 kernel
    (demo_gemm_split_plan / demo_gemm_set_split_equal).  The plan of the training step's N = 768 shape, and the results of the plans --
    one launch, main + remainder, equal parts -- on a shape with every epilogue operand that is offset per part (bf16 residual, second
    output as D8 codes, act' operand as D8 codes)."""
    import ctypes as C
    from demo2_amd import _lib
    from demo2_amd._lib import ACT_SAVED
    lib = _lib.load()
    # the plan itself (no launch): M = 49536, N = 768 -> 582 tiles = 2.27 rounds
    d = _lib.GemmDesc()
    x = torch.empty(16, device=DEV, dtype=torch.bfloat16)
    d.A = d.B = d.C = x.data_ptr()
    d.M, d.N, d.K = 49536, 768, 3072
    d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.c_rs = 3072, 1, 3072, 1, 768
    d.dtA = d.dtB = d.dtC = _lib.BF16
    d.alpha, d.batch1, d.batch2 = 1.0, 1, 1
    rows, cfgs = (C.c_int32 * 4)(), (C.c_int32 * 4)()
    lib.demo_gemm_set_split_equal(1024)                    # (the equal-parts plan is an opt-in hook: off by default)
    assert lib.demo_gemm_split_plan(C.byref(d), rows, cfgs, 4) == 2                                   # long K: main + remainder
    assert list(rows)[:2] == [170 * 256, 49536 - 170 * 256] and list(cfgs)[:2] == [2566, 1603]
    d.K = 768                                              # short K: three equal launches
    assert lib.demo_gemm_split_plan(C.byref(d), rows, cfgs, 4) == 3
    assert list(rows)[:3] == [65 * 256, 65 * 256, 49536 - 130 * 256] and list(cfgs)[:3] == [2566] * 3
    lib.demo_gemm_set_split_equal(0)
    assert lib.demo_gemm_split_plan(C.byref(d), rows, cfgs, 4) == 2
    M, N, K = 194 * 256 - 100, 768, 128
    a, b, bias = rnd(M, K).bfloat16().to(DEV), rnd(N, K, seed=1, scale=K ** -0.5).bfloat16().to(DEV), rnd(N, seed=2).to(DEV)
    res = rnd(M, N, seed=3).bfloat16().to(DEV)
    outs = []
    try:
        for split, eq in ((0, 0), (128, 0), (128, 4096)):
            lib.demo_gemm_set_split(split)
            lib.demo_gemm_set_split_equal(eq)
            y = ops.gemm(a, b, bias=bias, residual=res)
            u, codes = ops.gemm(a, b, bias=bias, act=ACT_QUICKGELU, pre_out_new=True, pre_out_deriv=True, pre_out_d8=True)
            g = ops.gemm(a, b, act_grad_pre=codes, act_grad_kind=ACT_SAVED)
            outs.append((y.float().cpu(), u.float().cpu(), codes.cpu(), g.float().cpu()))
    finally:
        lib.demo_gemm_set_split(128)
        lib.demo_gemm_set_split_equal(0)
    for o in outs[1:]:
        for t0, t1 in zip(outs[0], o):
            assert torch.equal(t0, t1)                     # same kernels' arithmetic on every plan (the bias is the accumulators' initial value in both MFMA kernels)
    ref = a.float().cpu() @ b.float().cpu().t() + bias.cpu() + res.float().cpu()
    torch.testing.assert_close(outs[0][0], ref, rtol=2e-2, atol=2e-2)


def test_gemm_bf16_batched_and_in_place_residual():
    Z, M, N, K = 3, 130, 128, 64
    A, B = rnd(Z, M, K).bfloat16(), rnd(Z, N, K, seed=1).bfloat16()
    Cd = torch.empty(Z, M, N, device=DEV, dtype=torch.float32)
    ops.gemm_raw(A.to(DEV), B.to(DEV), Cd, M, N, K, K, 1, K, 1, N, batch1=Z, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0))
    torch.testing.assert_close(Cd.cpu(), A.float() @ B.float().transpose(-1, -2), rtol=1e-3, atol=1e-3)
    x = rnd(M, N, seed=4).to(DEV)
    want = x.cpu() + A[0].float() @ B[0].float().t()
    ops.gemm(A[0].to(DEV), B[0].to(DEV), residual=x, out=x)
    torch.testing.assert_close(x.cpu(), want, rtol=1e-3, atol=1e-3)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,D", [(7, 128), (258, 768), (33, 1024), (5, 1536)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layernorm_fwd(M, D, dt):
    x, w, b = rnd(M, D, scale=2.0) + 0.5, 1 + rnd(D, seed=1, scale=0.1), rnd(D, seed=2, scale=0.1)
    want = O.layer_norm(x, w, b, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, dt, save=True)
    tol = dict(rtol=1e-5, atol=1e-5) if dt == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(y.float().cpu(), want, **tol)
    torch.testing.assert_close(mean.cpu(), x.mean(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rstd.cpu(), (x.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,D", [(300, 768), (9, 128), (2500, 768)])       # M >= 1024 takes the two-stage dw/db reduction
def test_layernorm_bwd(M, D):
    x = (rnd(M, D) + 0.3).requires_grad_(True)
    w, b = (1 + rnd(D, seed=1, scale=0.1)).requires_grad_(True), rnd(D, seed=2, scale=0.1).requires_grad_(True)
    dy = rnd(M, D, seed=3)
    O.layer_norm(x, w, b, 1e-5).backward(dy)
    _, mean, rstd = ops.layernorm_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), 1e-5, torch.float32, save=True)
    dw, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), mean, rstd, dw=dw, db=db)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-4)
    base = rnd(M, D, seed=7)
    dx2 = ops.layernorm_bwd(dy.bfloat16().to(DEV), x.detach().to(DEV), w.detach().to(DEV), mean, rstd, dx=base.to(DEV), accumulate_dx=True)
    torch.testing.assert_close(dx2.cpu(), base + x.grad, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float32])
def test_layernorm_add_fwd(xdt):
    """demo_layernorm_add_fwd: s = round_to_stream(x + add); xsum_out = s (also in place); y = LN(s) in bf16; statistics of s."""
    M, D = 259, 768
    x = (rnd(M, D, scale=2.0) + 0.5).to(xdt)
    add = rnd(M, D, seed=4).bfloat16()
    w, b = 1 + rnd(D, seed=1, scale=0.1), rnd(D, seed=2, scale=0.1)
    ssum = (x.float() + add.float()).to(xdt).float()
    want = O.layer_norm(ssum, w, b, 1e-5)
    xs = torch.empty(M, D, device=DEV, dtype=xdt)
    y, mean, rstd = ops.layernorm_add_fwd(x.to(DEV), add.to(DEV), w.to(DEV), b.to(DEV), 1e-5, xsum_out=xs, save=True)
    assert y.dtype == torch.bfloat16
    torch.testing.assert_close(xs.float().cpu(), ssum, rtol=0, atol=0)
    torch.testing.assert_close(y.float().cpu(), want, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean.cpu(), ssum.mean(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rstd.cpu(), (ssum.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5)
    xin = x.to(DEV).clone()                                      # in place, and without an addend
    y2 = ops.layernorm_add_fwd(xin, add.to(DEV), w.to(DEV), b.to(DEV), 1e-5, xsum_out=xin)
    assert torch.equal(xin, xs) and torch.equal(y2, y)
    y3 = ops.layernorm_add_fwd(x.to(DEV), None, w.to(DEV), b.to(DEV), 1e-5)
    torch.testing.assert_close(y3.float().cpu(), O.layer_norm(x.float(), w, b, 1e-5), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,D", [(300, 768), (2500, 768), (37, 128), (70, 512), (9000, 512)])
def test_layernorm_bwd_lp_stream(M, D):
    """demo_layernorm_bwd_lp: bf16(dres + LayerNorm-backward(dy)) with the running gradient in bf16, no fp32 stream; dw/db as usual;
    in place (output == running gradient) as the single-stream schedule uses it."""
    x = (rnd(M, D) + 0.3).requires_grad_(True)
    w, b = (1 + rnd(D, seed=1, scale=0.1)).requires_grad_(True), rnd(D, seed=2, scale=0.1).requires_grad_(True)
    dy = rnd(M, D, seed=3).bfloat16()
    O.layer_norm(x, w, b, 1e-5).backward(dy.float())
    _, mean, rstd = ops.layernorm_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), 1e-5, torch.float32, save=True)
    res = rnd(M, D, seed=7).bfloat16()
    dw, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    buf = res.to(DEV).clone()
    out = ops.layernorm_bwd_lp(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), mean, rstd, buf, buf, dw=dw, db=db)
    assert out.data_ptr() == buf.data_ptr() and out.dtype == torch.bfloat16
    want = (res.float() + x.grad).bfloat16().float()
    torch.testing.assert_close(out.float().cpu(), want, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-3)
    out2 = ops.layernorm_bwd_lp(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), mean, rstd, None, torch.empty(M, D, device=DEV, dtype=torch.bfloat16))
    torch.testing.assert_close(out2.float().cpu(), x.grad.bfloat16().float(), rtol=1e-2, atol=1e-2)
    # bf16 residual stream: x itself is bf16 (statistics taken from the rounded values) -- a wave per row with two rows ahead (1), half a wave per row (3), the 128-thread 8-column form (2), the 4-column one (0)
    from demo2_amd import _lib
    xb = (rnd(M, D, seed=11) + 0.3).bfloat16()
    xr = xb.float().requires_grad_(True)
    O.layer_norm(xr, w.detach(), b.detach(), 1e-5).backward(dy.float())
    _, mean, rstd = ops.layernorm_add_fwd(xb.to(DEV), None, w.detach().to(DEV), b.detach().to(DEV), 1e-5, save=True)
    wref = (dy.float() * ((xr.detach() - xr.detach().mean(1, keepdim=True)) * torch.rsqrt(xr.detach().var(1, unbiased=False, keepdim=True) + 1e-5))).sum(0)
    for lp8 in (1, 3, 2, 0):
        _lib.load().demo_layernorm_set_lp8(lp8)
        try:
            yv, mv, rv = ops.layernorm_add_fwd(xb.to(DEV), None, w.detach().to(DEV), b.detach().to(DEV), 1e-5, save=True)     # forward: half-wave / wave per row
            torch.testing.assert_close(yv.float().cpu(), O.layer_norm(xb.float(), w.detach(), b.detach(), 1e-5), rtol=1e-2, atol=1e-2)
            torch.testing.assert_close(mv.cpu(), xb.float().mean(1), rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(rv.cpu(), torch.rsqrt(xb.float().var(1, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
            dw, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
            buf = res.to(DEV).clone()
            out = ops.layernorm_bwd_lp(dy.to(DEV), xb.to(DEV), w.detach().to(DEV), mean, rstd, buf, buf, dw=dw, db=db)
        finally:
            _lib.load().demo_layernorm_set_lp8(1)
        torch.testing.assert_close(out.float().cpu(), (res.float() + xr.grad).bfloat16().float(), rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(dw.cpu(), wref, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(db.cpu(), dy.float().sum(0), rtol=1e-4, atol=1e-3)


def test_batchnorm_fwd_plain_and_grouped():
    Bn, Fd = 16, 96
    x, w, b = rnd(Bn, Fd) * 2 + 1, 1 + rnd(Fd, seed=1, scale=0.1), rnd(Fd, seed=2, scale=0.1)
    rm, rv = rnd(Fd, seed=3, scale=0.1), (1 + rnd(Fd, seed=4, scale=0.1)).abs()
    want = F.batch_norm(x, rm.clone(), rv.clone(), w, b, False, 0.1, 1e-5)
    got = ops.batchnorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV), False, 0.1, 1e-5)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
    rm2, rv2 = rm.clone(), rv.clone()
    want = F.batch_norm(x, rm2, rv2, w, b, True, 0.1, 1e-5)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    got, sm, si = ops.batchnorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), rmd, rvd, True, 0.1, 1e-5, save=True)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rmd.cpu(), rm2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rvd.cpu(), rv2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sm.cpu(), x.mean(0), rtol=1e-5, atol=1e-5)
    with pytest.raises(Exception):
        ops.batchnorm_fwd(x[:1].contiguous().to(DEV), w.to(DEV), b.to(DEV), rmd, rvd, True, 0.1, 1e-5)
    # grouped: feature f = j*C + h*d + k uses parameter block (h*7 + j), gate[b,h,j]
    H, d, J = 2, 8, 7
    Cc = H * d
    x = rnd(Bn, J * Cc) + 0.2
    pstride, sstride = 3 * d + 5, 2 * d + 3
    pw, pb = rnd(H * J * pstride, seed=5), rnd(H * J * pstride, seed=6)
    srm, srv = rnd(H * J * sstride, seed=7, scale=0.1), (1 + rnd(H * J * sstride, seed=8, scale=0.1)).abs()
    gate = torch.rand(Bn, H, J, generator=torch.Generator().manual_seed(3))
    want = torch.empty_like(x)
    for j in range(J):
        for h in range(H):
            blk = h * J + j
            sl = slice(j * Cc + h * d, j * Cc + (h + 1) * d)
            y = F.batch_norm(x[:, sl], srm[blk * sstride:blk * sstride + d], srv[blk * sstride:blk * sstride + d],
                             pw[blk * pstride:blk * pstride + d], pb[blk * pstride:blk * pstride + d], False, 0.1, 1e-5)
            want[:, sl] = y * gate[:, h, j, None]
    got = ops.batchnorm_fwd(x.to(DEV), pw.to(DEV), pb.to(DEV), srm.to(DEV), srv.to(DEV), False, 0.1, 1e-5,
                            group=(d, H, J, pstride, sstride), gate=gate.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, Bn, L, H):
    D = H * 64
    q, k, v = qkv.float().reshape(Bn, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    o = s.softmax(-1) @ v
    return o.transpose(1, 2).reshape(Bn * L, D), torch.logsumexp(s, -1).reshape(-1)


@pytest.mark.parametrize("Bn,L,H", [(3, 129, 12), (2, 9, 2), (1, 144, 3), (2, 17, 1), (2, 145, 2), (1, 160, 1), (5, 128, 4)])
@pytest.mark.parametrize("dt,variant", [(torch.bfloat16, 1), (torch.bfloat16, 0), (torch.float32, 1)])
def test_attention_fwd(Bn, L, H, dt, variant):
    """variant 1 (default): K/V-in-LDS form, 9 key tiles up to L = 144 and 10 up to 160; variant 0: the round-1 kernel."""
    from demo2_amd import _lib
    lib = _lib.load()
    qkv = rnd(Bn * L, 3 * H * 64, scale=1.5).to(dt)
    want, lse_want = attn_ref(qkv, Bn, L, H)
    lib.demo_attention_set_variant(variant)
    try:
        out, lse = ops.attention_fwd(qkv.to(DEV), Bn, L, H, save_lse=True)
    finally:
        lib.demo_attention_set_variant(1)
    tol = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out.float().cpu(), want, **tol)
    torch.testing.assert_close(lse.cpu(), lse_want, rtol=1e-3, atol=1e-3)


def test_attention_large_logits_and_generic_long():
    # a spiked key must not overflow the un-rescaled softmax (max is taken over the whole row first)
    Bn, L, H = 1, 129, 1
    qkv = rnd(L, 192).bfloat16()
    qkv[5, :64] = 30.0
    qkv[77, 64:128] = 30.0
    want, _ = attn_ref(qkv, Bn, L, H)
    out = ops.attention_fwd(qkv.to(DEV), Bn, L, H)
    assert torch.isfinite(out.float()).all()
    torch.testing.assert_close(out.float().cpu(), want, rtol=2e-2, atol=2e-2)
    qkv = rnd(200, 192).bfloat16()                     # L > 144 takes the generic kernel
    want, _ = attn_ref(qkv, 1, 200, 1)
    torch.testing.assert_close(ops.attention_fwd(qkv.to(DEV), 1, 200, 1).float().cpu(), want, rtol=2e-2, atol=2e-2)


# ------------------------------------------------------------------------------------------------ embedding / layout
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_im2col_and_embed_tokens(dt):
    Bn, H, W, Wd = 3, 64, 32, 128
    spec = O.Spec(img_h=H, img_w=W, width=Wd)
    img = rnd(Bn, 3, H, W)
    cols = ops.im2col(img.to(DEV), 16, 16, dt)
    want = O.patchify(img, spec).reshape(Bn * 8, 768)
    torch.testing.assert_close(cols.float().cpu(), want.to(dt).float(), rtol=0, atol=0)
    patch = rnd(Bn * 8, Wd, seed=1).to(dt)
    cls, pos, ce = rnd(Wd, seed=2), rnd(9, Wd, seed=3), rnd(4, Wd, seed=4)
    cam = torch.tensor([3, 0, 2])
    lw, lb = 1 + rnd(Wd, seed=5, scale=0.1), rnd(Wd, seed=6, scale=0.1)
    for all_tok in (False, True):
        for use_ln in (True, False):
            x = torch.cat([cls.expand(Bn, 1, Wd), patch.float().reshape(Bn, 8, Wd)], 1) + pos
            add = 1.5 * ce[cam][:, None, :]
            x = x + add if all_tok else torch.cat([x[:, :1] + add, x[:, 1:]], 1)
            want = O.layer_norm(x, lw, lb, 1e-5) if use_ln else x
            got, pre, mean, rstd = ops.embed_tokens(patch.to(DEV), cls.to(DEV), pos.to(DEV), ce.to(DEV), cam.to(DEV), 1.5, all_tok,
                                                    lw.to(DEV) if use_ln else None, lb.to(DEV) if use_ln else None, 1e-5, Bn, 8, save=True)
            torch.testing.assert_close(got.cpu().reshape(Bn, 9, Wd), want, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(pre.cpu().reshape(Bn, 9, Wd), x, rtol=1e-6, atol=1e-6)


def test_small_layout_kernels():
    x = rnd(1000, 7)
    assert torch.equal(ops.cast_f32_bf16(x.reshape(-1).to(DEV), torch.empty(7000, device=DEV, dtype=torch.bfloat16)).cpu(), x.reshape(-1).bfloat16())
    xb = x.bfloat16().reshape(-1)
    assert torch.equal(ops.cast_bf16_f32(xb.to(DEV), torch.empty(7000, device=DEV)).cpu(), xb.float())
    for dt in (torch.float32, torch.bfloat16):
        for shape in ((130, 70), (136, 72), (768, 2304), (200, 64)):      # multiples of 8 take the 16-B-access bf16 form, ragged tiles included
            t = rnd(*shape).to(dt)
            assert torch.equal(ops.transpose(t.to(DEV)).cpu(), t.t().contiguous())
    flat = rnd(3 * 5000).bfloat16().to(DEV)                                  # batched form on equally spaced matrices of a flat buffer
    dstb = torch.empty(3, 48, 72, device=DEV, dtype=torch.bfloat16)
    ops.transpose_batched(flat, 8, 5000, 72, 48, 3, dstb)
    for z in range(3):
        assert torch.equal(dstb[z].cpu(), flat[8 + z * 5000: 8 + z * 5000 + 72 * 48].view(72, 48).t().contiguous().cpu())
    t = rnd(777, 100)
    torch.testing.assert_close(ops.colsum(t.to(DEV)).cpu(), t.sum(0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ops.colsum(t.bfloat16().to(DEV)).cpu(), t.bfloat16().float().sum(0), rtol=1e-4, atol=1e-3)
    Bn, L, Cd = 4, 9, 64
    for dt in (torch.float32, torch.bfloat16):
        tok = rnd(Bn * L, Cd).to(dt)
        z = ops.pife_pool_cat(tok.to(DEV), Bn, L)
        tf = tok.float().reshape(Bn, L, Cd)
        torch.testing.assert_close(z.cpu(), torch.cat([tf[:, 0], tf[:, 1:].mean(1)], -1), rtol=1e-5, atol=1e-5)
        ori = torch.zeros(Bn, 3 * Cd, device=DEV)
        ops.get_token0(tok.to(DEV), ori, Cd, 3 * Cd, Bn, L)
        torch.testing.assert_close(ori.cpu()[:, Cd:2 * Cd], tf[:, 0], rtol=0, atol=0)
        glob = rnd(Bn, 3 * Cd, seed=2)
        td = tok.to(DEV).clone()
        ops.set_token0(td, glob.to(DEV), 2 * Cd, 3 * Cd, Bn, L)
        want = tf.clone()
        want[:, 0] = glob[:, 2 * Cd:].to(dt).float()
        torch.testing.assert_close(td.float().cpu().reshape(Bn, L, Cd), want, rtol=0, atol=0)


# ------------------------------------------------------------------------------------------------ HDM / ATMoE pieces
STREAMS = [(0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)]


@pytest.mark.parametrize("Bs,L,Cd,Hh,dt", [(3, 9, 128, 2, torch.float32), (2, 129, 512, 8, torch.bfloat16), (2, 5, 768, 12, torch.float32)])
def test_hdm_softmax_pool(Bs, L, Cd, Hh, dt):
    ldS = 128
    S = rnd(3 * Bs * L, ldS, scale=2.0)
    tok = rnd(3 * Bs * L, Cd, seed=1).to(dt)
    pooled, psum, probs = ops.hdm_softmax_pool(S.to(DEV), tok.to(DEV), Bs, L, Cd, Hh, ldS, save_probs=True)
    tf, Sf = tok.float().reshape(3, Bs, L, Cd), S.reshape(3, Bs, L, ldS)
    for br, st in enumerate(STREAMS):
        sc = torch.cat([Sf[m][:, :, br * Hh:(br + 1) * Hh] for m in st], 1)            # (Bs, Lk, Hh)
        x = torch.cat([tf[m] for m in st], 1)                                          # (Bs, Lk, C)
        p = sc.softmax(1)
        want = torch.einsum("blh,blc->bhc", p, x)
        torch.testing.assert_close(pooled[:, br].cpu(), want, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(probs[:, br, :, :len(st) * L].cpu(), p.transpose(1, 2), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(psum.cpu(), torch.ones(Bs, 7, Hh), rtol=1e-5, atol=1e-5)


def test_hdm_dropout_is_a_scaled_mask():
    Bs, L, Cd, Hh, ldS = 4, 129, 128, 2, 128
    S, tok = rnd(3 * Bs * L, ldS), rnd(3 * Bs * L, Cd, seed=1)
    _, _, p0 = ops.hdm_softmax_pool(S.to(DEV), tok.to(DEV), Bs, L, Cd, Hh, ldS, save_probs=True)
    pooled, psum, ppre, p1 = ops.hdm_softmax_pool(S.to(DEV), tok.to(DEV), Bs, L, Cd, Hh, ldS, p_drop=0.1, seed=1234, save_probs=True,
                                                  save_probs_post=True)
    _, _, _, p2 = ops.hdm_softmax_pool(S.to(DEV), tok.to(DEV), Bs, L, Cd, Hh, ldS, p_drop=0.1, seed=1234, save_probs=True,
                                       save_probs_post=True)
    assert torch.equal(p1, p2)                                               # counter-based: reproducible for backward
    assert torch.equal(ppre, p0)                                             # the saved softmax is the pre-dropout one
    tf = tok.reshape(3, Bs, L, Cd)
    for br, st in enumerate(STREAMS):                                        # pooled uses the post-dropout probabilities
        x = torch.cat([tf[m] for m in st], 1)
        want = torch.einsum("bhl,blc->bhc", p1[:, br, :, :len(st) * L].cpu(), x)
        torch.testing.assert_close(pooled[:, br].cpu(), want, rtol=1e-4, atol=1e-4)
    kept = p1 != 0
    valid = p0 > 0
    frac = 1 - kept[valid].float().mean().item()
    assert 0.08 < frac < 0.12
    torch.testing.assert_close(p1[kept], (p0 / 0.9)[kept], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(psum, p1.sum(-1), rtol=1e-5, atol=1e-6)
    y = torch.zeros(Bs, 7, Cd, device=DEV)
    bias = rnd(7 * 200, seed=3)
    ops.hdm_vbias(y, bias.to(DEV), 10, 200, psum, Bs, Cd, Hh)
    want = torch.stack([bias[10 + br * 200: 10 + br * 200 + Cd] for br in range(7)])[None] * psum.cpu().repeat_interleave(Cd // Hh, -1)
    torch.testing.assert_close(y.cpu(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("Bs,Cd,H", [(5, 128, 4), (3, 512, 8), (2, 768, 12)])
def test_atmoe_gate_softmax(Bs, Cd, H):
    q, k = rnd(Bs, Cd), rnd(Bs * 7, Cd, seed=1)
    d = Cd // H
    want = (torch.einsum("bhd,bjhd->bhj", q.reshape(Bs, H, d), k.reshape(Bs, 7, H, d)) * d ** -0.5).softmax(-1)
    got = ops.atmoe_gate_softmax(q.to(DEV), k.to(DEV), Bs, Cd, H)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------ fusion module vs the reference's golden outputs
@pytest.mark.parametrize("name,C,H", [("fusion_c128_h4", 128, 4), ("fusion_c64_h2", 64, 2)])
@pytest.mark.parametrize("fused", [True, False])
def test_general_fusion_eval_against_reference_golden(golden, name, C, H, fused):
    """GeneralFusion (HDM + ATMoE) at inference on the HIP schedule against the outputs of the reference's OWN module
    (tests/golden/fusion_*.npz, oracle/gen_golden.py: gen_fusion) -- HDM features, gate tensor and the (B, 7C) mixture, B in {1, 2, 4} --
    with ATMoE in its fused two-launch form (csrc/atmoe.hip) and in the multi-launch form training uses; the weight-only quantities
    of HDM come from the per-parameter-version cache on the second and third batch."""
    from demo2_amd.modeling import hip_forward as HF, make_model
    from util import small_cfg
    g = golden(name)
    cfg = small_cfg("clip", head=H, TRUNK_OUT_DIM=C)
    torch.manual_seed(0)
    m = make_model(cfg, 11, 3, 0)
    missing = m.load_state_dict({"generalFusion." + k: v for k, v in g["sd"].items()}, strict=False)
    assert not [k for k in missing.unexpected_keys] and not [k for k in missing.missing_keys if k.startswith("generalFusion.")]
    m = m.to(DEV).eval().set_compute_dtype(torch.float32)
    m.fused_atmoe_eval = fused
    if fused:
        assert ops.atmoe_eval_supported(C, H)
    for B in (1, 2, 4):
        e = g[f"eval_b{B}"]
        L = e["in0"].shape[1] + 1
        # token streams as the trunk leaves them: (3B*L, C), row 0 of every image = the (PIFE) global, rows 1.. = the patch tokens
        tokens = torch.cat([torch.cat([e[f"in{3 + i}"][:, None, :], e[f"in{i}"]], 1) for i in range(3)], 0).reshape(3 * B * L, C)
        ori = torch.cat([e[f"in{3 + i}"] for i in range(3)], 1)
        m.geom.n_patches = L - 1
        feats = HF.hdm_forward(m, tokens.to(DEV), ori.to(DEV), B, False)
        for i in range(7):
            torch.testing.assert_close(feats[:, i].cpu(), e[f"hdm{i}"].reshape(B, C), rtol=2e-4, atol=2e-5)
        moe = HF.atmoe_forward(m, feats, B, False)
        torch.testing.assert_close(moe.cpu(), e["out"].reshape(B, 7 * C), rtol=2e-4, atol=2e-5)
        if fused:                                               # the gate tensor itself (AttnMOE.py:61-68), from the fused kernels
            P, S, gk = m.ptab, m.stab, "generalFusion.moe.gating_network.gate."
            e0, e1 = "generalFusion.moe.experts.0.expertHead.0.mlp.", "generalFusion.moe.experts.0.expertHead.1.mlp."
            bn1 = (P.f(gk + "linear_re.2.weight"), P.f(gk + "linear_re.2.bias"), S.f(gk + "linear_re.2.running_mean"), S.f(gk + "linear_re.2.running_var"))
            gfold = ops.atmoe_fold_keys(P.f(gk + "q_.weight"), P.f(gk + "k_.weight"), H)
            _, gates = ops.atmoe_eval_fwd(feats, P.f(gk + "linear_re.0.weight"), P.f(gk + "linear_re.0.bias"), bn1, gfold, P.f(e0 + "0.weight"), P.f(e0 + "0.bias"), P.f(e0 + "2.weight"), P.f(e0 + "2.bias"),
                                          S.f(e0 + "2.running_mean"), S.f(e0 + "2.running_var"), P.off(e1 + "0.weight") - P.off(e0 + "0.weight"),
                                          S.off(e1 + "2.running_mean") - S.off(e0 + "2.running_mean"), 1e-5, B, C, H, want_gates=True)
            torch.testing.assert_close(gates.cpu(), e["gate"].reshape(B, H, 7), rtol=2e-4, atol=2e-5)
    assert m.weight_cache("hdm_qU", torch.float32) is not None
    # a parameter update invalidates the cache
    with torch.no_grad():
        m.generalFusion.r_token.add_(0.5)
    assert m.weight_cache("hdm_qU", torch.float32) is None


@pytest.mark.parametrize("B,C,H", [(1, 512, 4), (37, 512, 8), (130, 768, 12)])
def test_atmoe_eval_fused_equals_multi_launch_at_full_width(B, C, H):
    """Configs A / B / V widths (d = 128 / 64 / 64), ragged sample blocks: the two-launch form against the multi-launch schedule."""
    from demo2_amd.modeling import hip_forward as HF, make_model
    from util import make_cfg
    kw = dict(HEAD=H) if C == 512 else dict(HEAD=H, TRANSFORMER_TYPE="vit_base_patch16_224", SIE_COE=3.0, TRUNK_WIDTH=768)
    cfg = make_cfg("RGBNT201", TRUNK_LAYERS=1, **kw)
    torch.manual_seed(0)
    m = make_model(cfg, 11, 3, 0)
    sd = O.init_state_dict(__import__("util").spec_of(m), seed=3)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    feats = rnd(B, 7, C, seed=B).to(DEV)
    m.fused_atmoe_eval = True
    a = HF.atmoe_forward(m, feats, B, False)
    m.fused_atmoe_eval = False
    b = HF.atmoe_forward(m, feats, B, False)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
