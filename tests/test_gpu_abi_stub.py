"""GPU: the stand-alone ctypes binding INTEGRATION.md shows a maintainer (no demo2_amd imports: the documented struct layout,
argument order and error protocol are exactly what the library exports) -- c_fc Linear + QuickGELU of one CLIP block
(reference modeling/clip/model.py:173-177) against torch fp32."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "demo2_amd", "lib", "libdemo_hip.so")


class GemmDesc(C.Structure):                      # struct demo_gemm_desc (include/demo_hip.h), as printed in INTEGRATION.md
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("a_rs", C.c_int64), ("a_cs", C.c_int64), ("b_rs", C.c_int64), ("b_cs", C.c_int64), ("c_rs", C.c_int64),
                ("dtA", C.c_int32), ("dtB", C.c_int32), ("dtC", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float),
                ("batch1", C.c_int32), ("batch2", C.c_int32),
                ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
                ("sC1", C.c_int64), ("sC2", C.c_int64), ("sBias1", C.c_int64), ("sBias2", C.c_int64),
                ("act_grad_pre", C.c_void_p), ("dtPre", C.c_int32), ("act_grad_kind", C.c_int32), ("accumulate", C.c_int32),
                ("pre_out", C.c_void_p), ("dtRes", C.c_int32), ("pre_out_kind", C.c_int32)]


def test_documented_ctypes_stub_runs_the_mlp_linear():
    lib = C.CDLL(LIB)
    assert lib.demo_gemm_desc_size() == C.sizeof(GemmDesc) and lib.demo_abi_version() == 1
    lib.demo_gemm.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    lib.demo_last_error.restype = C.c_char_p
    M, N, K = 3 * 16 * 129, 3072, 768
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = GemmDesc(A=x.data_ptr(), B=W.data_ptr(), C=y.data_ptr(), bias=b.data_ptr(), M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=K, b_cs=1,
                 c_rs=N, dtA=1, dtB=1, dtC=1, act=1, alpha=1.0, batch1=1, batch2=1)
    rc = lib.demo_gemm(C.byref(d), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.demo_last_error()
    torch.cuda.synchronize()
    pre = x.float() @ W.float().t() + b
    want = pre * torch.sigmoid(1.702 * pre)
    torch.testing.assert_close(y.float(), want, rtol=2e-2, atol=2e-2)
    # error protocol: a bad descriptor returns a negative code and a message, nothing is launched
    bad = GemmDesc(A=x.data_ptr(), B=W.data_ptr(), C=None, M=M, N=N, K=K, a_rs=K, a_cs=1, b_rs=K, b_cs=1, c_rs=N, dtA=1, dtB=1, dtC=1,
                   alpha=1.0, batch1=1, batch2=1)
    assert lib.demo_gemm(C.byref(bad), None) < 0 and b"null" in lib.demo_last_error()
