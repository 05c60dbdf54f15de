"""Thin Python launchers over the C ABI (include/demo_hip.h).  PyTorch is used only for device
memory and streams: every function here hands raw device pointers + the current HIP stream to
libdemo_hip.so.  Tensors must live on a HIP device; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_SAVED, BF16, D8, F32, GemmDesc, check

Tensor = torch.Tensor

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt_code(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"demo2_amd supports float32/bfloat16 tensors, got {t}")


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DemoHipError("demo2_amd ops need HIP device tensors (no CPU fallback)")


class DeviceErrors:
    """Caller-owned device error word of the C ABI (include/demo_hip.h, DEMO_ERR_*): kernels that take an index argument (camera
    ids, class labels) cannot raise the way torch's indexing does in the reference, so they clamp / poison the offending sample and OR a
    bit into this word.  It is read back WITHOUT a device synchronisation: `post()` queues an async copy into pinned host memory
    behind the kernels issued so far, `poll()` (called at the start of the next forward) raises once that copy has completed and
    shows a bit; `check()` synchronises and raises (tests, end of an evaluation, before a checkpoint is written)."""
    MESSAGES = {_lib.ERR_CAM_RANGE: "cam_label outside [0, camera_num) (the reference's cv_embed[cam_label] raises IndexError)",
                _lib.ERR_LABEL_RANGE: "class label outside [0, num_classes) (the reference's label-smoothing scatter_ raises)"}

    def __init__(self, device):
        self.word = torch.zeros(1, device=device, dtype=torch.int32)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.word.is_cuda else None
        self.event = None

    def ptr(self):
        return self.word.data_ptr()

    def post(self):
        if self.host is None or torch.cuda.is_current_stream_capturing():
            return
        self.host.copy_(self.word, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def _raise_if(self, bits: int):
        if bits:
            self.word.zero_()
            if self.host is not None:
                self.host.zero_()
            self.event = None
            what = "; ".join(m for b, m in self.MESSAGES.items() if bits & b) or f"error bits {bits:#x}"
            raise IndexError("demo2_amd kernel reported an out-of-range index in an earlier launch: " + what)

    def poll(self):
        if self.event is not None and self.event.query():
            self.event = None
            self._raise_if(int(self.host[0]))

    def check(self):
        self._raise_if(int(self.word.item()))


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def gemm_raw(A: Tensor, B: Tensor, Cout: Tensor, M: int, N: int, K: int, a_rs: int, a_cs: int, b_rs: int, b_cs: int,
             c_rs: int, *, a_off: int = 0, b_off: int = 0, c_off: int = 0, bias: Optional[Tensor] = None, bias_off: int = 0,
             residual: Optional[Tensor] = None, act: int = ACT_NONE, alpha: float = 1.0, batch1: int = 1, batch2: int = 1,
             sA=(0, 0), sB=(0, 0), sC=(0, 0), sBias=(0, 0), act_grad_pre: Optional[Tensor] = None, act_grad_kind: int = 0,
             accumulate: bool = False, pre_out: Optional[Tensor] = None, pre_out_deriv: bool = False):
    """Fully explicit descriptor launch; offsets are in elements of the respective tensor."""
    _req_cuda(A, B, Cout, bias, residual, act_grad_pre)
    d = GemmDesc()
    d.A = A.data_ptr() + a_off * A.element_size()
    d.B = B.data_ptr() + b_off * B.element_size()
    d.C = Cout.data_ptr() + c_off * Cout.element_size()
    d.bias = None if bias is None else bias.data_ptr() + bias_off * 4
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be fp32")
    if residual is not None:
        d.residual = residual.data_ptr() + c_off * residual.element_size()
        d.dtRes = dt_code(residual.dtype)                 # fp32, or the bf16 residual stream of the bf16 compute mode
    else:
        d.residual = None
        d.dtRes = 0
    d.M, d.N, d.K = M, N, K
    d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.c_rs = a_rs, a_cs, b_rs, b_cs, c_rs
    d.dtA, d.dtB, d.dtC = dt_code(A.dtype), dt_code(B.dtype), dt_code(Cout.dtype)
    d.act, d.alpha = act, alpha
    d.batch1, d.batch2 = batch1, batch2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.sBias1, d.sBias2 = sBias
    if act_grad_pre is not None:
        d.act_grad_pre = act_grad_pre.data_ptr() + c_off * act_grad_pre.element_size()
        if act_grad_pre.dtype == torch.uint8:             # DEMO_D8 codes of a derivative an earlier forward saved (pre_out_deriv, uint8 pre_out)
            if act_grad_kind != ACT_SAVED:
                raise TypeError("a uint8 act_grad_pre holds DEMO_D8 codes of a saved derivative: act_grad_kind must be ACT_SAVED")
            d.dtPre = D8
        else:
            d.dtPre = dt_code(act_grad_pre.dtype)
        d.act_grad_kind = act_grad_kind
    else:
        d.act_grad_pre = None
        d.dtPre = 0
        d.act_grad_kind = 0
    d.accumulate = 1 if accumulate else 0
    d.pre_out_kind = 0
    if pre_out is not None:
        if pre_out.dtype == torch.uint8:                  # act'(x) as DEMO_D8 codes, one byte per element
            if not pre_out_deriv:
                raise TypeError("a uint8 pre_out receives the DEMO_D8 codes of act'(x): pass pre_out_deriv=True")
            d.pre_out_kind = 2
        elif pre_out.dtype != Cout.dtype:
            raise TypeError("pre_out must have the dtype of C (or uint8 with pre_out_deriv: DEMO_D8 codes)")
        else:
            d.pre_out_kind = 1 if pre_out_deriv else 0
        d.pre_out = pre_out.data_ptr() + c_off * pre_out.element_size()
    else:
        d.pre_out = None
    check(_lib.load().demo_gemm(C.byref(d), _stream()), "demo_gemm")
    return Cout


def gemm(a: Tensor, b: Tensor, *, out: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
         bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, act: int = ACT_NONE, alpha: float = 1.0,
         act_grad_pre: Optional[Tensor] = None, act_grad_kind: int = 0, accumulate: bool = False,
         pre_out: Optional[Tensor] = None, pre_out_new: bool = False, pre_out_deriv: bool = False, pre_out_d8: bool = False):
    """out[M,N] = act(alpha * a[M,K] @ b[N,K]^T + bias) (* act'(pre)) (+ residual).  a, b may be any 2-D strided views.
    pre_out receives the pre-activation, or its activation DERIVATIVE with pre_out_deriv (then the backward product takes
    act_grad_kind=ACT_SAVED: one multiply instead of re-evaluating act' from the pre-activation).  A uint8 pre_out (pre_out_new with
    pre_out_d8) receives the derivative as DEMO_D8 codes (demo_hip.h: one byte per element, |error| <= 0.0025); the backward product
    takes that uint8 tensor as act_grad_pre."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1], (a.shape, b.shape)
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or a.dtype)
    assert out.shape == (M, N) and out.stride(1) == 1
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride() == out.stride()
    if act_grad_pre is not None:
        assert act_grad_pre.shape == (M, N) and act_grad_pre.stride() == out.stride()
    if pre_out_new:
        pre_out = torch.empty_like(out, dtype=torch.uint8) if (pre_out_d8 and pre_out_deriv) else torch.empty_like(out)
    if pre_out is not None:
        assert pre_out.shape == (M, N) and pre_out.stride() == out.stride()
    gemm_raw(a, b, out, M, N, K, a.stride(0), a.stride(1), b.stride(0), b.stride(1), out.stride(0), bias=bias,
             residual=residual, act=act, alpha=alpha, act_grad_pre=act_grad_pre, act_grad_kind=act_grad_kind,
             accumulate=accumulate, pre_out=pre_out, pre_out_deriv=pre_out_deriv)
    return (out, pre_out) if pre_out_new else out


# ------------------------------------------------------------------------------------------------
# elementwise / layout
# ------------------------------------------------------------------------------------------------
def cast_f32_bf16(src: Tensor, dst: Tensor):
    _req_cuda(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    assert src.is_contiguous() and dst.is_contiguous()
    check(_lib.load().demo_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "demo_cast_f32_bf16")
    return dst


def cast_bf16_f32(src: Tensor, dst: Tensor):
    _req_cuda(src, dst)
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src.numel() == dst.numel()
    check(_lib.load().demo_cast_bf16_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "demo_cast_bf16_f32")
    return dst


def droppath_mask(n: int, keep_prob: float, seed: int, device, seed_dev: Optional[Tensor] = None) -> Tensor:
    """(n,) fp32: 0 or 1/keep_prob per image (vit_pytorch.py:48-65), from the library's counter RNG."""
    out = torch.empty(n, device=device, dtype=torch.float32)
    _req_cuda(out)
    check(_lib.load().demo_droppath_mask(out.data_ptr(), n, float(keep_prob), seed & (2 ** 64 - 1), _ptr(seed_dev), _stream()), "demo_droppath_mask")
    return out


def rowscale_add(x: Tensor, t: Tensor, scale: Tensor, Bn: int, L: int, out: Optional[Tensor] = None) -> Tensor:
    """out = x + scale[image] * t   (x, out (Bn*L, W) fp32; t compute dtype)"""
    _req_cuda(x, t, scale)
    W = x.shape[1]
    assert x.dtype == torch.float32 and x.is_contiguous() and t.is_contiguous() and t.shape == x.shape == (Bn * L, W) and scale.numel() == Bn
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().demo_rowscale_add(x.data_ptr(), t.data_ptr(), dt_code(t.dtype), scale.data_ptr(), out.data_ptr(), Bn, L, W, _stream()),
          "demo_rowscale_add")
    return out


def rowscale_cast(x: Tensor, scale: Tensor, Bn: int, L: int, dtype: torch.dtype) -> Tensor:
    """out = scale[image] * x   ((Bn*L, W) fp32 -> dtype)"""
    _req_cuda(x, scale)
    W = x.shape[1]
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape == (Bn * L, W) and scale.numel() == Bn
    out = torch.empty(Bn * L, W, device=x.device, dtype=dtype)
    check(_lib.load().demo_rowscale_cast(x.data_ptr(), scale.data_ptr(), out.data_ptr(), dt_code(dtype), Bn, L, W, _stream()), "demo_rowscale_cast")
    return out


def transpose(src: Tensor, dst: Optional[Tensor] = None) -> Tensor:
    _req_cuda(src)
    assert src.dim() == 2 and src.is_contiguous()
    r, c = src.shape
    if dst is None:
        dst = torch.empty(c, r, device=src.device, dtype=src.dtype)
    check(_lib.load().demo_transpose(src.data_ptr(), dst.data_ptr(), r, c, dt_code(src.dtype), _stream()), "demo_transpose")
    return dst


def transpose_batched(src_flat: Tensor, src_off: int, src_stride: int, rows: int, cols: int, batch: int, dst: Tensor) -> Tensor:
    """dst[z] (cols, rows) = src_flat[src_off + z*src_stride :][:rows*cols].view(rows, cols)^T for z < batch, one launch."""
    _req_cuda(src_flat, dst)
    assert dst.is_contiguous() and dst.shape == (batch, cols, rows) and dst.dtype == src_flat.dtype
    assert src_off + (batch - 1) * src_stride + rows * cols <= src_flat.numel()
    esz = src_flat.element_size()
    check(_lib.load().demo_transpose_batched(src_flat.data_ptr() + src_off * esz, dst.data_ptr(), rows, cols, batch, src_stride,
                                             rows * cols, dt_code(src_flat.dtype), _stream()), "demo_transpose_batched")
    return dst


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    _req_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
        accumulate = False
    check(_lib.load().demo_colsum(x.data_ptr(), out.data_ptr(), M, N, x.stride(0), dt_code(x.dtype), int(accumulate), _stream()),
          "demo_colsum")
    return out


# ------------------------------------------------------------------------------------------------
# patch embedding / tokens
# ------------------------------------------------------------------------------------------------
def im2col(img: Tensor, patch: int, stride: int, dtype: torch.dtype) -> Tensor:
    _req_cuda(img)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4 and img.shape[1] == 3
    Bn, _, H, W = img.shape
    ny, nx = (H - patch) // stride + 1, (W - patch) // stride + 1
    cols = torch.empty(Bn * ny * nx, 3 * patch * patch, device=img.device, dtype=dtype)
    check(_lib.load().demo_im2col(img.data_ptr(), cols.data_ptr(), Bn, H, W, patch, stride, dt_code(dtype), _stream()), "demo_im2col")
    return cols


def im2col_into(img: Tensor, cols: Tensor, patch: int, stride: int) -> Tensor:
    _req_cuda(img, cols)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4 and img.shape[1] == 3
    Bn, _, H, W = img.shape
    ny, nx = (H - patch) // stride + 1, (W - patch) // stride + 1
    assert cols.is_contiguous() and cols.shape == (Bn * ny * nx, 3 * patch * patch)
    check(_lib.load().demo_im2col(img.data_ptr(), cols.data_ptr(), Bn, H, W, patch, stride, dt_code(cols.dtype), _stream()), "demo_im2col")
    return cols


def embed_tokens(patch_out: Tensor, cls: Tensor, pos: Tensor, cam_embed: Optional[Tensor], cam: Optional[Tensor],
                 cam_scale: float, cam_all_tokens: bool, ln_w: Optional[Tensor], ln_b: Optional[Tensor], eps: float,
                 Bn: int, n_patch: int, save: bool = False, err: Optional[DeviceErrors] = None):
    """-> x (Bn*(n_patch+1), W) fp32 [, pre, mean, rstd when save].  cam_embed (n_cam, W); camera ids outside the table are clamped and
    reported through `err` (DeviceErrors)."""
    _req_cuda(patch_out, cls, pos)
    W = patch_out.shape[1]
    rows = Bn * (n_patch + 1)
    x = torch.empty(rows, W, device=patch_out.device, dtype=torch.float32)
    pre = mean = rstd = None
    if save:
        pre = torch.empty_like(x)
        if ln_w is not None:
            mean = torch.empty(rows, device=x.device, dtype=torch.float32)
            rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    if cam is not None:
        assert cam.dtype == torch.int64 and cam.numel() == Bn
    check(_lib.load().demo_embed_tokens(patch_out.data_ptr(), dt_code(patch_out.dtype), cls.data_ptr(), pos.data_ptr(),
                                        _ptr(cam_embed), _ptr(cam), 0 if cam_embed is None else cam_embed.shape[0], float(cam_scale),
                                        int(cam_all_tokens), _ptr(ln_w), _ptr(ln_b), float(eps), x.data_ptr(), _ptr(pre), _ptr(mean),
                                        _ptr(rstd), Bn, n_patch, W, None if err is None else err.ptr(), _stream()),
          "demo_embed_tokens")
    return (x, pre, mean, rstd) if save else x


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, eps: float, out_dtype: torch.dtype, save: bool = False,
                  out: Optional[Tensor] = None):
    _req_cuda(x, w, b)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    M, D = x.shape
    y = out if out is not None else torch.empty(M, D, device=x.device, dtype=out_dtype)
    assert y.is_contiguous() and y.shape == (M, D) and y.dtype == out_dtype
    mean = rstd = None
    if save:
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    check(_lib.load().demo_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps), y.data_ptr(), dt_code(out_dtype),
                                         _ptr(mean), _ptr(rstd), M, D, _stream()), "demo_layernorm_fwd")
    return (y, mean, rstd) if save else y


def layernorm_bwd(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dx: Optional[Tensor] = None,
                  accumulate_dx: bool = False, dw: Optional[Tensor] = None, db: Optional[Tensor] = None,
                  dx_lp: Optional[Tensor] = None):
    """dx (fp32); dw/db are ACCUMULATED into the given fp32 buffers (zero them first if needed);
    dx_lp (optional bf16 tensor of x's shape) receives a bf16 copy of the final dx."""
    _req_cuda(dy, x)
    M, D = x.shape
    if dx is None:
        dx = torch.empty_like(x)
        accumulate_dx = False
    if dx_lp is not None:
        assert dx_lp.dtype == torch.bfloat16 and dx_lp.is_contiguous() and dx_lp.shape == x.shape
    ws, ws_n = None, 0
    if M >= 1024 and (dw is not None or db is not None):        # partial-row workspace (one per device/shape, reused)
        lib = _lib.load()
        ws_n = int(lib.demo_layernorm_bwd_workspace(M, D))
        key = (x.device, ws_n)
        ws = _LN_WS.get(key)
        if ws is None:
            ws = _LN_WS[key] = torch.empty(ws_n, device=x.device, dtype=torch.float32)
    check(_lib.load().demo_layernorm_bwd(dy.data_ptr(), dt_code(dy.dtype), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         dx.data_ptr(), int(accumulate_dx), _ptr(dx_lp), _ptr(dw), _ptr(db), _ptr(ws), ws_n, M, D, _stream()),
          "demo_layernorm_bwd")
    return dx


def layernorm_add_fwd(x: Tensor, add: Optional[Tensor], w: Tensor, b: Tensor, eps: float, xsum_out: Optional[Tensor] = None, save: bool = False):
    """s = x (+ add), rounded to x's dtype; xsum_out (optional, may be x itself) = s; -> y = LN(s) in bf16 [, mean, rstd when save]"""
    _req_cuda(x, add, w, b)
    assert x.dim() == 2 and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    M, D = x.shape
    if add is not None:
        assert add.dtype == torch.bfloat16 and add.is_contiguous() and add.shape == x.shape
    if xsum_out is not None:
        assert xsum_out.dtype == x.dtype and xsum_out.is_contiguous() and xsum_out.shape == x.shape
    y = torch.empty(M, D, device=x.device, dtype=torch.bfloat16)
    mean = rstd = None
    if save:
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    check(_lib.load().demo_layernorm_add_fwd(x.data_ptr(), dt_code(x.dtype), _ptr(add), BF16, _ptr(xsum_out), w.data_ptr(), b.data_ptr(), float(eps),
                                             y.data_ptr(), BF16, _ptr(mean), _ptr(rstd), M, D, _stream()), "demo_layernorm_add_fwd")
    return (y, mean, rstd) if save else y


def layernorm_bwd_lp(dy: Tensor, x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dres_lp: Optional[Tensor], out_lp: Tensor,
                     dw: Optional[Tensor] = None, db: Optional[Tensor] = None) -> Tensor:
    """out_lp (bf16) = bf16(dres_lp (bf16, optional) + LayerNorm-backward(dy)); dw/db accumulated as in layernorm_bwd."""
    _req_cuda(dy, x, out_lp)
    M, D = x.shape
    assert out_lp.dtype == torch.bfloat16 and out_lp.is_contiguous() and out_lp.shape == x.shape and dy.is_contiguous() and dy.shape == x.shape
    if dres_lp is not None:
        assert dres_lp.dtype == torch.bfloat16 and dres_lp.is_contiguous() and dres_lp.shape == x.shape
    ws, ws_n = None, 0
    lib = _lib.load()
    if M >= 1024 and (dw is not None or db is not None):
        ws_n = int(lib.demo_layernorm_bwd_workspace(M, D))
        key = (x.device, ws_n)
        ws = _LN_WS.get(key)
        if ws is None:
            ws = _LN_WS[key] = torch.empty(ws_n, device=x.device, dtype=torch.float32)
    check(lib.demo_layernorm_bwd_lp(dy.data_ptr(), dt_code(dy.dtype), x.data_ptr(), dt_code(x.dtype), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres_lp),
                                    out_lp.data_ptr(), _ptr(dw), _ptr(db), _ptr(ws), ws_n, M, D, _stream()), "demo_layernorm_bwd_lp")
    return out_lp


_LN_WS = {}
_TN_WS = {}        # per device: partial-tile workspace of the two-pass split-K weight-gradient GEMM


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, shadow: Optional[Tensor], lr: float, beta1: float, beta2: float, eps: float,
              weight_decay: float, step: int, zero_grad: bool = True, hyper_dev: Optional[Tensor] = None):
    """Fused flat Adam (torch.optim.Adam semantics, L2 weight decay folded into the gradient) over the flat parameter /
    gradient buffers; optionally refreshes the bf16 weight shadow and zeroes the gradient in the same pass.
    hyper_dev: optional fp32 CUDA tensor {lr, update count t, -} read on the device instead of lr / step."""
    _req_cuda(p, g, m, v)
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and (shadow is None or shadow.numel() == n)
    assert hyper_dev is None or (hyper_dev.dtype == torch.float32 and hyper_dev.numel() == 3 and hyper_dev.is_cuda)
    check(_lib.load().demo_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(shadow), n, float(lr), float(beta1),
                                     float(beta2), float(eps), float(weight_decay), int(step), int(zero_grad), _ptr(hyper_dev), _stream()),
          "demo_adam_step")


OPT_ADAM, OPT_ADAMW, OPT_SGD = 0, 1, 2


def optim_step(kind: int, p: Tensor, g: Tensor, m: Tensor, v: Optional[Tensor], shadow: Optional[Tensor], lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float, step: int, zero_grad: bool = True, hyper_dev: Optional[Tensor] = None,
               block_class: Optional[Tensor] = None, class_table: Optional[Tensor] = None):
    """Flat optimizer step in general form (demo_optim_step): kind OPT_ADAM / OPT_ADAMW / OPT_SGD (beta1 = momentum); block_class
    (uint8 per 64 elements) + class_table (fp32 {lr multiplier, weight decay} per class) are the reference's parameter groups."""
    _req_cuda(p, g, m, v, block_class, class_table)
    n = p.numel()
    assert g.numel() == n and m.numel() == n and (v is None or v.numel() == n) and (shadow is None or shadow.numel() == n)
    if block_class is not None:
        assert block_class.dtype == torch.uint8 and block_class.numel() * 64 == n and class_table.dtype == torch.float32
        assert class_table.numel() % 2 == 0 and block_class.is_contiguous() and class_table.is_contiguous()
    check(_lib.load().demo_optim_step(int(kind), p.data_ptr(), g.data_ptr(), m.data_ptr(), _ptr(v), _ptr(shadow), n, float(lr), float(beta1),
                                      float(beta2), float(eps), float(weight_decay), int(step), int(zero_grad), _ptr(hyper_dev),
                                      _ptr(block_class), _ptr(class_table), _stream()), "demo_optim_step")


def batchnorm_fwd(x: Tensor, w: Tensor, b: Tensor, running_mean: Tensor, running_var: Tensor, training: bool, momentum: float,
                  eps: float, out_dtype: torch.dtype = torch.float32, save: bool = False, group=None, gate: Optional[Tensor] = None):
    """group = (gsz, n_inner, n_outer, pstride, sstride) for the grouped-parameter form (see header)."""
    _req_cuda(x, w, b, running_mean, running_var)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    Bn, Fd = x.shape
    y = torch.empty(Bn, Fd, device=x.device, dtype=out_dtype)
    sm = si = None
    if save and training:
        sm = torch.empty(Fd, device=x.device, dtype=torch.float32)
        si = torch.empty(Fd, device=x.device, dtype=torch.float32)
    gsz, ni, no, ps, ss = group if group is not None else (0, 1, 1, 0, 0)
    check(_lib.load().demo_batchnorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                         int(training), float(momentum), float(eps), y.data_ptr(), dt_code(out_dtype), _ptr(sm), _ptr(si),
                                         Bn, Fd, gsz, ni, no, ps, ss, _ptr(gate), _stream()), "demo_batchnorm_fwd")
    return (y, sm, si) if save else y


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attention_fwd(qkv: Tensor, Bn: int, L: int, H: int, save_lse: bool = False):
    _req_cuda(qkv)
    D = H * 64
    assert qkv.shape == (Bn * L, 3 * D) and qkv.is_contiguous()
    out = torch.empty(Bn * L, D, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(Bn * H * L, device=qkv.device, dtype=torch.float32) if save_lse else None
    check(_lib.load().demo_attention_fwd(qkv.data_ptr(), out.data_ptr(), _ptr(lse), Bn, L, H, 64, dt_code(qkv.dtype), _stream()),
          "demo_attention_fwd")
    return (out, lse) if save_lse else out


# ------------------------------------------------------------------------------------------------
# PIFE / HDM / ATMoE pieces
# ------------------------------------------------------------------------------------------------
def pife_pool_cat(tokens: Tensor, Bn: int, L: int) -> Tensor:
    _req_cuda(tokens)
    Cd = tokens.shape[-1]
    assert tokens.is_contiguous() and tokens.numel() == Bn * L * Cd
    z = torch.empty(Bn, 2 * Cd, device=tokens.device, dtype=torch.float32)
    check(_lib.load().demo_pife_pool_cat(tokens.data_ptr(), dt_code(tokens.dtype), z.data_ptr(), Bn, L, Cd, _stream()), "demo_pife_pool_cat")
    return z


def set_token0(tokens: Tensor, glob: Tensor, glob_off: int, glob_ld: int, Bn: int, L: int):
    """tokens[b,0,:] = glob_flat[glob_off + b*glob_ld : ... + C]"""
    _req_cuda(tokens, glob)
    Cd = tokens.shape[-1]
    assert glob.dtype == torch.float32 and tokens.is_contiguous() and tokens.numel() == Bn * L * Cd
    check(_lib.load().demo_set_token0(tokens.data_ptr(), dt_code(tokens.dtype), glob.data_ptr() + 4 * glob_off, glob_ld, Bn, L, Cd, _stream()),
          "demo_set_token0")
    return tokens


def get_token0(tokens: Tensor, out: Tensor, out_off: int, out_ld: int, Bn: int, L: int):
    """out_flat[out_off + b*out_ld : ... + C] = tokens[b,0,:] (fp32)"""
    _req_cuda(tokens, out)
    Cd = tokens.shape[-1]
    assert out.dtype == torch.float32 and tokens.is_contiguous() and tokens.numel() == Bn * L * Cd
    check(_lib.load().demo_get_token0(tokens.data_ptr(), dt_code(tokens.dtype), out.data_ptr() + 4 * out_off, out_ld, Bn, L, Cd, _stream()),
          "demo_get_token0")
    return out


def hdm_softmax_pool(S: Tensor, tokens: Tensor, Bs: int, L: int, Cd: int, Hh: int, ldS: int, p_drop: float = 0.0, seed: int = 0,
                     save_probs: bool = False, save_probs_post: bool = False, seed_dev: Optional[Tensor] = None):
    """-> pooled (Bs,7,Hh,C), psum (Bs,7,Hh), probs (softmax BEFORE dropout) [, probs_post (after dropout)]
    seed_dev: optional 1-element int64 CUDA tensor added to `seed` on the device (graph-captured training steps)."""
    _req_cuda(S, tokens)
    assert S.dtype == torch.float32 and S.shape == (3 * Bs * L, ldS) and S.is_contiguous() and ldS >= 7 * Hh
    assert tokens.is_contiguous() and tokens.numel() == 3 * Bs * L * Cd
    pooled = torch.empty(Bs, 7, Hh, Cd, device=S.device, dtype=torch.float32)
    psum = torch.empty(Bs, 7, Hh, device=S.device, dtype=torch.float32)
    probs = torch.empty(Bs, 7, Hh, 3 * L, device=S.device, dtype=torch.float32) if save_probs else None
    post = torch.empty(Bs, 7, Hh, 3 * L, device=S.device, dtype=torch.float32) if save_probs_post else None
    check(_lib.load().demo_hdm_softmax_pool(S.data_ptr(), ldS, tokens.data_ptr(), dt_code(tokens.dtype), pooled.data_ptr(), psum.data_ptr(),
                                            _ptr(probs), _ptr(post), Bs, L, Cd, Hh, float(p_drop), int(seed) & (2 ** 64 - 1), _ptr(seed_dev),
                                            _stream()), "demo_hdm_softmax_pool")
    return (pooled, psum, probs, post) if save_probs_post else (pooled, psum, probs)


def hdm_softmax_pool_bwd(dpooled: Tensor, dpsum: Optional[Tensor], probs: Tensor, tokens: Tensor, pooled: Tensor, psum: Tensor,
                         Bs: int, L: int, Cd: int, Hh: int, ldS: int, p_drop: float = 0.0, seed: int = 0,
                         seed_dev: Optional[Tensor] = None):
    """-> dS (3Bs*L, ldS) fp32, dtokens (3Bs*L, C) fp32"""
    _req_cuda(dpooled, probs, tokens, pooled, psum)
    dS = torch.empty(3 * Bs * L, ldS, device=tokens.device, dtype=torch.float32)
    dtok = torch.empty(3 * Bs * L, Cd, device=tokens.device, dtype=torch.float32)
    check(_lib.load().demo_hdm_softmax_pool_bwd(dpooled.data_ptr(), _ptr(dpsum), probs.data_ptr(), tokens.data_ptr(), dt_code(tokens.dtype),
                                                pooled.data_ptr(), psum.data_ptr(), dS.data_ptr(), ldS, dtok.data_ptr(), Bs, L, Cd, Hh,
                                                float(p_drop), int(seed) & (2 ** 64 - 1), _ptr(seed_dev), _stream()),
          "demo_hdm_softmax_pool_bwd")
    return dS, dtok


def hdm_vbias_bwd(dy: Tensor, bias_base: Tensor, bias_off: int, bias_stride: int, psum: Tensor, dbias_base: Tensor, dpsum: Tensor,
                  Bs: int, Cd: int, Hh: int):
    _req_cuda(dy, bias_base, psum, dbias_base, dpsum)
    check(_lib.load().demo_hdm_vbias_bwd(dy.data_ptr(), bias_base.data_ptr() + 4 * bias_off, bias_stride, psum.data_ptr(),
                                         dbias_base.data_ptr() + 4 * bias_off, dpsum.data_ptr(), Bs, Cd, Hh, _stream()), "demo_hdm_vbias_bwd")


def atmoe_gate_softmax_bwd(dgates: Tensor, gates: Tensor, q: Tensor, k: Tensor, Bs: int, Cd: int, H: int):
    _req_cuda(dgates, gates, q, k)
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    check(_lib.load().demo_atmoe_gate_softmax_bwd(dgates.data_ptr(), gates.data_ptr(), q.data_ptr(), k.data_ptr(), dq.data_ptr(),
                                                  dk.data_ptr(), Bs, Cd, H, _stream()), "demo_atmoe_gate_softmax_bwd")
    return dq, dk


# ------------------------------------------------------------------------------------------------
# DGAF-V3 (dual_gated_fusion.py:558-745)
# ------------------------------------------------------------------------------------------------
def dgaf_attn_pool(S: Tensor, tokens: Tensor, Bs: int, L: int, Cd: int, nh: int, ldS: int, save_probs: bool = False):
    """-> pooled (Bs,3,nh,C) fp32, probs (Bs,3,nh,L-1) fp32 or None"""
    _req_cuda(S, tokens)
    assert S.dtype == torch.float32 and S.is_contiguous() and S.shape == (3 * Bs * L, ldS)
    assert tokens.is_contiguous() and tokens.shape == (3 * Bs * L, Cd)
    pooled = torch.empty(Bs, 3, nh, Cd, device=S.device, dtype=torch.float32)
    probs = torch.empty(Bs, 3, nh, L - 1, device=S.device, dtype=torch.float32) if save_probs else None
    check(_lib.load().demo_dgaf_attn_pool(S.data_ptr(), ldS, tokens.data_ptr(), dt_code(tokens.dtype), pooled.data_ptr(), _ptr(probs), Bs, L, Cd,
                                          nh, _stream()), "demo_dgaf_attn_pool")
    return pooled, probs


def dgaf_attn_pool_bwd(dpooled: Tensor, pooled: Tensor, probs: Tensor, tokens: Tensor, Bs: int, L: int, Cd: int, nh: int, ldS: int):
    """-> dS (3*Bs*L, ldS) fp32, dtokens (3*Bs*L, C) fp32 (both fully written)"""
    _req_cuda(dpooled, pooled, probs, tokens)
    for t, shp in ((dpooled, (Bs, 3, nh, Cd)), (pooled, (Bs, 3, nh, Cd)), (probs, (Bs, 3, nh, L - 1))):
        assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shp, (t.shape, shp)
    assert tokens.is_contiguous() and tokens.shape == (3 * Bs * L, Cd)
    dS = torch.empty(3 * Bs * L, ldS, device=tokens.device, dtype=torch.float32)
    dtok = torch.empty(3 * Bs * L, Cd, device=tokens.device, dtype=torch.float32)
    check(_lib.load().demo_dgaf_attn_pool_bwd(dpooled.data_ptr(), pooled.data_ptr(), probs.data_ptr(), tokens.data_ptr(), dt_code(tokens.dtype),
                                              dS.data_ptr(), ldS, dtok.data_ptr(), Bs, L, Cd, nh, _stream()), "demo_dgaf_attn_pool_bwd")
    return dS, dtok


def dgaf_gate_fuse(h: Tensor, E: Tensor, g1n: Tensor, W2: Tensor, b2: Tensor, alpha_raw: Tensor, tau: float, save: bool = False):
    """h, E (Bs,3,C) ; g1n (Bs,C) -> fused (Bs,C) [, relu(g1n) (Bs,C), aux (Bs,16) when save]"""
    _req_cuda(h, E, g1n, W2, b2, alpha_raw)
    Bs, Cd = g1n.shape
    for t, n in ((h, 3 * Bs * Cd), (E, 3 * Bs * Cd), (g1n, Bs * Cd), (W2, 3 * Cd), (b2, 3), (alpha_raw, 1)):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    fused = torch.empty(Bs, Cd, device=h.device, dtype=torch.float32)
    relu = torch.empty(Bs, Cd, device=h.device, dtype=torch.float32) if save else None
    aux = torch.empty(Bs, 16, device=h.device, dtype=torch.float32) if save else None
    check(_lib.load().demo_dgaf_gate_fuse(h.data_ptr(), E.data_ptr(), g1n.data_ptr(), W2.data_ptr(), b2.data_ptr(), alpha_raw.data_ptr(),
                                          float(tau), fused.data_ptr(), _ptr(relu), _ptr(aux), Bs, Cd, _stream()), "demo_dgaf_gate_fuse")
    return (fused, relu, aux) if save else fused


def dgaf_gate_fuse_bwd(dfused: Tensor, dres: Optional[Tensor], h: Tensor, g1n: Tensor, W2: Tensor, aux: Tensor, alpha_raw: Tensor, tau: float):
    """-> dh (Bs,3,C), dE (Bs,3,C), dg1n (Bs,C), dlogit (Bs,3), dalpha (Bs)"""
    _req_cuda(dfused, dres, h, g1n, W2, aux, alpha_raw)
    Bs, Cd = g1n.shape
    for t, n in ((dfused, Bs * Cd), (h, 3 * Bs * Cd), (g1n, Bs * Cd), (W2, 3 * Cd), (aux, 16 * Bs), (alpha_raw, 1)):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    if dres is not None:
        assert dres.dtype == torch.float32 and dres.is_contiguous() and dres.numel() == 3 * Bs * Cd
    dev = h.device
    dh = torch.empty(Bs, 3, Cd, device=dev, dtype=torch.float32)
    dE = torch.empty(Bs, 3, Cd, device=dev, dtype=torch.float32)
    dg1n = torch.empty(Bs, Cd, device=dev, dtype=torch.float32)
    dlogit = torch.empty(Bs, 3, device=dev, dtype=torch.float32)
    dalpha = torch.empty(Bs, device=dev, dtype=torch.float32)
    check(_lib.load().demo_dgaf_gate_fuse_bwd(dfused.data_ptr(), _ptr(dres), h.data_ptr(), g1n.data_ptr(), W2.data_ptr(), aux.data_ptr(),
                                              alpha_raw.data_ptr(), float(tau), dh.data_ptr(), dE.data_ptr(), dg1n.data_ptr(), dlogit.data_ptr(),
                                              dalpha.data_ptr(), Bs, Cd, _stream()), "demo_dgaf_gate_fuse_bwd")
    return dh, dE, dg1n, dlogit, dalpha


def dgaf_combine(h: Tensor, enh: Tensor) -> Tensor:
    """out (Bs, 3C): out[b, s*C + c] = h[b,s,c] + enh[b,c]"""
    _req_cuda(h, enh)
    Bs, Cd = enh.shape
    assert h.dtype == torch.float32 and enh.dtype == torch.float32 and h.is_contiguous() and enh.is_contiguous() and h.numel() == 3 * Bs * Cd
    out = torch.empty(Bs, 3 * Cd, device=h.device, dtype=torch.float32)
    check(_lib.load().demo_dgaf_combine(h.data_ptr(), enh.data_ptr(), out.data_ptr(), Bs, Cd, _stream()), "demo_dgaf_combine")
    return out


def dgaf_sum3(dout: Tensor, Cd: int) -> Tensor:
    """denh (Bs, C) = sum_s dout[:, s*C:(s+1)*C]"""
    _req_cuda(dout)
    Bs = dout.shape[0]
    assert dout.dtype == torch.float32 and dout.is_contiguous() and dout.shape == (Bs, 3 * Cd)
    denh = torch.empty(Bs, Cd, device=dout.device, dtype=torch.float32)
    check(_lib.load().demo_dgaf_sum3(dout.data_ptr(), denh.data_ptr(), Bs, Cd, _stream()), "demo_dgaf_sum3")
    return denh


def gemm_tn(a: Tensor, b: Tensor, out: Tensor, out_off: int = 0, ldc: Optional[int] = None, alpha: float = 1.0, accumulate: bool = True,
            colsum_out: Optional[Tensor] = None, colsum_off: int = 0):
    """out_flat[out_off + m*ldc + n] (+)= alpha * sum_r a[r, m] * b[r, n]   (weight gradients: a = dY, b = X).
    bf16 operands with 128-multiples take the MFMA kernel; everything else the generic strided kernel.
    colsum_out (optional): fp32 buffer that also receives += sum_r a[r, :] (the bias gradient), fused into the MFMA kernel."""
    _req_cuda(a, b, out)
    R, Mo = a.shape
    No = b.shape[1]
    assert b.shape[0] == R and out.dtype == torch.float32
    ldc = No if ldc is None else ldc
    if (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and Mo % 128 == 0 and No % 128 == 0 and a.stride(1) == 1
            and b.stride(1) == 1 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        cs = None if colsum_out is None else colsum_out.data_ptr() + 4 * colsum_off
        lib = _lib.load()
        # partial-tile workspace of the two-pass split-K form (one per device and stream, grown to the largest request: <= 66 MB)
        need = int(lib.demo_gemm_tn_workspace(Mo, No, R))
        wkey = (a.device, _stream())                 # per stream: concurrent launches on two streams must not share it
        ws = _TN_WS.get(wkey)
        if need and (ws is None or ws.numel() < need):
            ws = _TN_WS[wkey] = torch.empty(need, device=a.device, dtype=torch.float32)
        check(lib.demo_gemm_tn_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr() + 4 * out_off, Mo, No, R, a.stride(0), b.stride(0), ldc,
                                    float(alpha), int(accumulate), cs, _ptr(ws) if need else None, ws.numel() if (need and ws is not None) else 0,
                                    _stream()), "demo_gemm_tn_bf16")
        return out
    if colsum_out is not None:                      # bias gradient: colsum_out[colsum_off + m] += sum_r a[r, m]
        colsum(a, out=colsum_out[colsum_off:colsum_off + Mo], accumulate=True)
    return gemm_raw(a, b, out, Mo, No, R, a.stride(1), a.stride(0), b.stride(1), b.stride(0), ldc, c_off=out_off, alpha=alpha,
                    accumulate=accumulate)


def gemm_tn_group(items, out: Tensor, accumulate: bool = True, alpha: float = 1.0):
    """Up to four weight-gradient products over the same reduction rows as ONE launch (demo_gemm_tn_bf16_group):
    items = [(a (R, Mo), b (R, No), out_off, ldc, colsum_off or None), ...]; out_flat[out_off + m*ldc + n] (+)= sum_r a[r, m] b[r, n] and,
    with colsum_off, out_flat[colsum_off + m] += sum_r a[r, m] (bias gradient).  Falls back to one gemm_tn per item when the group's
    contract (bf16, Mo / No multiples of 256, R >= 2048, 16-B alignment, colsum for all or none) does not hold."""
    lib = _lib.load()
    R = items[0][0].shape[0]
    ok = 1 <= len(items) <= 4 and R >= 2048 and out.dtype == torch.float32
    cs_flags = [it[4] is not None for it in items]
    ok = ok and (all(cs_flags) or not any(cs_flags))
    for a, b, _, _, _ in items:
        ok = ok and (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape[0] == R and b.shape[0] == R and a.shape[1] % 256 == 0
                     and b.shape[1] % 256 == 0 and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
                     and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)
    if not ok:
        for a, b, out_off, ldc, cs_off in items:
            gemm_tn(a, b, out, out_off=out_off, ldc=ldc, alpha=alpha, accumulate=accumulate, colsum_out=out if cs_off is not None else None,
                    colsum_off=cs_off or 0)
        return out
    _req_cuda(out, *[t for it in items for t in it[:2]])
    arr = (_lib.TnItem * len(items))()
    for i, (a, b, out_off, ldc, cs_off) in enumerate(items):
        arr[i].A, arr[i].B, arr[i].C = a.data_ptr(), b.data_ptr(), out.data_ptr() + 4 * out_off
        arr[i].Mo, arr[i].No = a.shape[1], b.shape[1]
        arr[i].lda, arr[i].ldb, arr[i].ldc = a.stride(0), b.stride(0), ldc
        arr[i].colsum = None if cs_off is None else out.data_ptr() + 4 * cs_off
    need = int(lib.demo_gemm_tn_group_workspace(arr, len(items), R))
    wkey = (out.device, _stream(), "group")
    ws = _TN_WS.get(wkey)
    if ws is None or ws.numel() < need:
        ws = _TN_WS[wkey] = torch.empty(need, device=out.device, dtype=torch.float32)
    check(lib.demo_gemm_tn_bf16_group(arr, len(items), R, float(alpha), int(accumulate), ws.data_ptr(), ws.numel(), _stream()),
          "demo_gemm_tn_bf16_group")
    return out


def attention_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, Bn: int, L: int, H: int) -> Tensor:
    _req_cuda(qkv, out, dout, lse)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dout.dtype == qkv.dtype
    dqkv = torch.empty_like(qkv)
    check(_lib.load().demo_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), Bn, L, H, 64,
                                         dt_code(qkv.dtype), _stream()), "demo_attention_bwd")
    return dqkv


def embed_tokens_bwd(dpre: Tensor, cam: Optional[Tensor], cam_scale: float, cam_all_tokens: bool, dpatch_dtype: torch.dtype,
                     dcls: Tensor, dpos: Tensor, dcam: Optional[Tensor], Bn: int, n_patch: int, n_cam: int = 0,
                     err: Optional[DeviceErrors] = None) -> Tensor:
    """dcls / dpos / dcam are accumulated into (fp32 views of the flat gradient buffer; dcam has n_cam rows); returns dpatch (Bn*n_patch, W)."""
    if dcam is not None and cam is not None and n_cam <= 0:
        raise ValueError("embed_tokens_bwd: n_cam (rows of the camera table) is required with dcam")
    _req_cuda(dpre, dcls, dpos)
    W = dpre.shape[1]
    dpatch = torch.empty(Bn * n_patch, W, device=dpre.device, dtype=dpatch_dtype)
    check(_lib.load().demo_embed_tokens_bwd(dpre.data_ptr(), _ptr(cam), int(n_cam), float(cam_scale), int(cam_all_tokens), dpatch.data_ptr(),
                                            dt_code(dpatch_dtype), dcls.data_ptr(), dpos.data_ptr(), _ptr(dcam), Bn, n_patch, W,
                                            None if err is None else err.ptr(), _stream()),
          "demo_embed_tokens_bwd")
    return dpatch


def act_fwd(pre: Tensor, kind: int, out: Optional[Tensor] = None) -> Tensor:
    _req_cuda(pre)
    assert pre.dtype == torch.float32 and pre.is_contiguous()
    out = torch.empty_like(pre) if out is None else out
    check(_lib.load().demo_act_fwd(pre.data_ptr(), kind, out.data_ptr(), pre.numel(), _stream()), "demo_act_fwd")
    return out


def act_bwd(dy: Tensor, pre: Tensor, kind: int, out: Optional[Tensor] = None) -> Tensor:
    _req_cuda(dy, pre)
    assert dy.dtype == torch.float32 and pre.dtype == torch.float32 and dy.is_contiguous() and pre.is_contiguous()
    out = torch.empty_like(dy) if out is None else out
    check(_lib.load().demo_act_bwd(dy.data_ptr(), pre.data_ptr(), kind, out.data_ptr(), dy.numel(), _stream()), "demo_act_bwd")
    return out


def batchnorm_bwd(dy: Tensor, x: Tensor, w: Tensor, b: Optional[Tensor], save_mean: Tensor, save_invstd: Tensor, dw: Optional[Tensor],
                  db: Optional[Tensor], dx: Optional[Tensor] = None, accumulate_dx: bool = False, group=None, gate: Optional[Tensor] = None,
                  dgate: Optional[Tensor] = None) -> Tensor:
    """dw/db: fp32 tensors whose data_ptr is the (group) base; accumulated into.  group = (gsz, n_inner, n_outer, pstride)."""
    _req_cuda(dy, x, w)
    Bn, Fd = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dy.dtype == torch.float32
    if dx is None:
        dx, accumulate_dx = torch.empty_like(x), False
    gsz, ni, no, ps = group if group is not None else (0, 1, 1, 0)
    check(_lib.load().demo_batchnorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _ptr(b), save_mean.data_ptr(), save_invstd.data_ptr(),
                                         dx.data_ptr(), int(accumulate_dx), _ptr(dw), _ptr(db), Bn, Fd, gsz, ni, no, ps, _ptr(gate), _ptr(dgate),
                                         _stream()), "demo_batchnorm_bwd")
    return dx


def token0_grad(dtokens: Tensor, out: Tensor, out_off: int, out_ld: int, Bn: int, L: int):
    _req_cuda(dtokens, out)
    Cd = dtokens.shape[-1]
    check(_lib.load().demo_token0_grad(dtokens.data_ptr(), out.data_ptr() + 4 * out_off, out_ld, Bn, L, Cd, _stream()), "demo_token0_grad")


def tokens_grad(dtok: Optional[Tensor], dcls: Optional[Tensor], cls_off: int, cls_ld: int, dmean: Optional[Tensor], mean_off: int,
                mean_ld: int, keep_row0: bool, out: Tensor, Bn: int, L: int):
    _req_cuda(out)
    Cd = out.shape[-1]
    pc = None if dcls is None else dcls.data_ptr() + 4 * cls_off
    pm = None if dmean is None else dmean.data_ptr() + 4 * mean_off
    check(_lib.load().demo_tokens_grad(_ptr(dtok), pc, cls_ld, pm, mean_ld, int(keep_row0), out.data_ptr(), dt_code(out.dtype), Bn, L, Cd,
                                       _stream()), "demo_tokens_grad")
    return out


def hdm_vbias(y: Tensor, bias_base: Tensor, bias_off: int, bias_stride: int, psum: Tensor, Bs: int, Cd: int, Hh: int):
    _req_cuda(y, bias_base, psum)
    check(_lib.load().demo_hdm_vbias(y.data_ptr(), bias_base.data_ptr() + 4 * bias_off, bias_stride, psum.data_ptr(), Bs, Cd, Hh, _stream()),
          "demo_hdm_vbias")
    return y


def atmoe_gate_softmax(q: Tensor, k: Tensor, Bs: int, Cd: int, H: int) -> Tensor:
    _req_cuda(q, k)
    assert q.dtype == torch.float32 and k.dtype == torch.float32 and q.is_contiguous() and k.is_contiguous()
    gates = torch.empty(Bs, H, 7, device=q.device, dtype=torch.float32)
    check(_lib.load().demo_atmoe_gate_softmax(q.data_ptr(), k.data_ptr(), gates.data_ptr(), Bs, Cd, H, _stream()), "demo_atmoe_gate_softmax")
    return gates


def atmoe_eval_supported(Cd: int, H: int) -> bool:
    return bool(_lib.load().demo_atmoe_eval_supported(int(Cd), int(H)))


def atmoe_fold_keys(wq: Tensor, wk: Tensor, H: int, out: Optional[Tensor] = None) -> Tensor:
    """gfold (H*C, C) = [Wq[h]^T Wk[h] for h < H] (Wq[h], Wk[h]: rows h*d..(h+1)*d of the gate's q_ / k_ weights): the weights-only
    factor of the gate scores, x2 . (gfold_h f_j) = (Wq[h] x2) . (Wk[h] f_j).  H small products on the library's GEMM; the caller
    caches the result per parameter version."""
    C = wq.shape[0]
    d = C // H
    assert wq.shape == wk.shape == (C, C) and C % H == 0
    out = torch.empty(H * C, C, device=wq.device, dtype=torch.float32) if out is None else out
    for h in range(H):
        gemm(wq[h * d:(h + 1) * d].t(), wk[h * d:(h + 1) * d].t(), out=out[h * C:(h + 1) * C])
    return out


def atmoe_eval_fwd(feats: Tensor, w1: Tensor, b1: Tensor, bn1: tuple, gfold: Tensor, ew: Tensor, eb: Tensor, ebn_w: Tensor,
                   ebn_b: Tensor, ebn_mean: Tensor, ebn_var: Tensor, expert_stride: int, stat_stride: int, eps: float, Bs: int, Cd: int,
                   H: int, ws: Optional[dict] = None, want_gates: bool = False):
    """ATMoE in eval mode, two launches (demo_atmoe_eval_fwd): feats (Bs,7,C) fp32 -> moe (Bs,7C) [, gates (Bs,H,7)].  bn1 = (weight,
    bias, running_mean, running_var) of the gate's BatchNorm; gfold = atmoe_fold_keys(q_.weight, k_.weight, H); e* tensors START at
    expert (0,0)'s parameter (views into the flat buffers), consecutive experts `expert_stride` (parameters) / `stat_stride` (running
    statistics) floats apart.  ws: a dict the caller keeps between calls (the three workspaces); None = allocate for this call."""
    _req_cuda(feats, w1, b1, gfold, ew, eb, ebn_w, ebn_b, ebn_mean, ebn_var, *bn1)
    assert feats.dtype == torch.float32 and feats.is_contiguous() and feats.numel() == Bs * 7 * Cd
    assert gfold.dtype == torch.float32 and gfold.is_contiguous() and gfold.shape == (H * Cd, Cd)
    dev = feats.device
    if ws is None:
        ws = {}
    if ws.get("B") != Bs or ws.get("C") != Cd or ws.get("H") != H or ws["y"].device != dev:
        ws.clear()
        ws.update(B=Bs, C=Cd, H=H, part=torch.empty(_lib.load().demo_atmoe_eval_part_elems(Bs, Cd), device=dev),
                  tkey=torch.empty(7 * Bs, H * Cd, device=dev), y=torch.empty(Bs, 7 * Cd, device=dev))
    moe = torch.empty(Bs, 7 * Cd, device=dev, dtype=torch.float32)
    gates = torch.empty(Bs, H, 7, device=dev, dtype=torch.float32) if want_gates else None
    check(_lib.load().demo_atmoe_eval_fwd(feats.data_ptr(), w1.data_ptr(), b1.data_ptr(), bn1[0].data_ptr(), bn1[1].data_ptr(), bn1[2].data_ptr(),
                                          bn1[3].data_ptr(), gfold.data_ptr(), ew.data_ptr(), eb.data_ptr(), ebn_w.data_ptr(),
                                          ebn_b.data_ptr(), ebn_mean.data_ptr(), ebn_var.data_ptr(), int(expert_stride), int(stat_stride),
                                          float(eps), ws["part"].data_ptr(), ws["tkey"].data_ptr(), ws["y"].data_ptr(), moe.data_ptr(),
                                          _ptr(gates), Bs, Cd, H, _stream()), "demo_atmoe_eval_fwd")
    return (moe, gates) if want_gates else moe


# ------------------------------------------------------------------------------------------------
# training losses (value + gradient in one launch)
# ------------------------------------------------------------------------------------------------
def xent_label_smooth(logits: Tensor, target: Tensor, loss: Tensor, eps: float = 0.1, weight: float = 1.0, need_grad: bool = True,
                      err: Optional[DeviceErrors] = None):
    """loss[0] += weight * CrossEntropyLabelSmooth(logits, target); returns weight * d/dlogits (or None).  A label outside [0, K) poisons
    that sample with NaN and is reported through `err`."""
    _req_cuda(logits, target, loss)
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1 and target.dtype == torch.int64
    assert loss.dtype == torch.float32 and loss.numel() >= 1
    B, K = logits.shape
    d = torch.empty(B, K, device=logits.device, dtype=torch.float32) if need_grad else None
    check(_lib.load().demo_xent_label_smooth(logits.data_ptr(), logits.stride(0), target.data_ptr(), B, K, float(eps), float(weight),
                                             loss.data_ptr(), _ptr(d), K, None if err is None else err.ptr(), _stream()), "demo_xent_label_smooth")
    return d


def triplet_soft_margin(feat: Tensor, labels: Tensor, loss: Tensor, weight: float = 1.0, need_grad: bool = True, margin: float = -1.0):
    """loss[0] += weight * TripletLoss(margin)(feat, labels) (batch-hard; margin < 0 = the soft-margin form, MODEL.NO_MARGIN True);
    returns weight * d/dfeat (or None)."""
    _req_cuda(feat, labels, loss)
    assert feat.dtype == torch.float32 and feat.dim() == 2 and feat.stride(1) == 1 and labels.dtype == torch.int64
    B, D = feat.shape
    d = torch.empty(B, D, device=feat.device, dtype=torch.float32) if need_grad else None
    ws = torch.empty(2 * B * B, device=feat.device, dtype=torch.float32)        # Gram matrix + gradient weights
    check(_lib.load().demo_triplet_loss(feat.data_ptr(), feat.stride(0), labels.data_ptr(), B, D, float(weight), float(margin), loss.data_ptr(),
                                        _ptr(d), D, ws.data_ptr(), _stream()), "demo_triplet_loss")
    return d


# ------------------------------------------------------------------------------------------------
# retrieval evaluator
# ------------------------------------------------------------------------------------------------
def l2_normalize_rows(x: Tensor, eps: float = 1e-12) -> Tensor:
    """In place: x[i] /= max(||x[i]||_2, eps)  (torch.nn.functional.normalize(dim=1))."""
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    check(_lib.load().demo_l2_normalize_rows(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], float(eps), _stream()), "demo_l2_normalize_rows")
    return x


def row_sqnorm(x: Tensor) -> Tensor:
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
    check(_lib.load().demo_row_sqnorm(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), _stream()), "demo_row_sqnorm")
    return out


def add_outer_sum(dist: Tensor, a: Tensor, b: Tensor) -> Tensor:
    """In place: dist[i, j] = (a[i] + b[j]) + dist[i, j]."""
    _req_cuda(dist, a, b)
    nq, ng = dist.shape
    assert dist.dtype == torch.float32 and dist.stride(1) == 1 and a.numel() == nq and b.numel() == ng
    check(_lib.load().demo_add_outer_sum(dist.data_ptr(), dist.stride(0), a.data_ptr(), b.data_ptr(), nq, ng, _stream()), "demo_add_outer_sum")
    return dist


def rank_eval(dist: Tensor, q_pids: Tensor, g_pids: Tensor, q_keys: Tensor, g_keys: Tensor, want_order: bool = False):
    """-> (ap (nq,) float64 or -1, first_hit (nq,) int32 or -1 [, order (nq, ng) int32]); see demo_rank_eval in the header."""
    _req_cuda(dist, q_pids, g_pids, q_keys, g_keys)
    nq, ng = dist.shape
    assert dist.dtype == torch.float32 and dist.stride(1) == 1
    for t_, n_ in ((q_pids, nq), (g_pids, ng), (q_keys, nq), (g_keys, ng)):
        assert t_.dtype == torch.int64 and t_.numel() == n_ and t_.is_contiguous()
    ap = torch.empty(nq, device=dist.device, dtype=torch.float64)
    first = torch.empty(nq, device=dist.device, dtype=torch.int32)
    order = torch.empty(nq, ng, device=dist.device, dtype=torch.int32) if want_order else None
    check(_lib.load().demo_rank_eval(dist.data_ptr(), dist.stride(0), q_pids.data_ptr(), g_pids.data_ptr(), q_keys.data_ptr(),
                                     g_keys.data_ptr(), nq, ng, ap.data_ptr(), first.data_ptr(), None, _ptr(order), _stream()),
          "demo_rank_eval")
    return (ap, first, order) if want_order else (ap, first)
