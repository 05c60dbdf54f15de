"""ctypes binding of libdemo_hip.so (the C ABI declared in include/demo_hip.h).

There is deliberately no fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using the product path raises.  Build it with
`make -C demo2_amd/csrc` (or `python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEMO_HIP_LIB: an alternative build of the same library (A/B runs of compile-time variants, e.g. make VARIANT=st2 EXTRA=-DDEMO_STORE_POL=2)
LIB_PATH = os.environ.get("DEMO_HIP_LIB") or os.path.join(_HERE, "lib", "libdemo_hip.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "demo_hip.h")

F32, BF16 = 0, 1
D8 = 2                # demo_hip.h DEMO_D8: 8-bit code of a saved activation derivative (uint8 tensors; demo_gemm's pre_out / act_grad_pre only)
ERR_CAM_RANGE, ERR_LABEL_RANGE = 1, 2          # include/demo_hip.h: bits of the device error word
ACT_NONE, ACT_QUICKGELU, ACT_GELU, ACT_SAVED = 0, 1, 2, 3      # ACT_SAVED: only as act_grad_kind (the operand is the derivative)


class DemoHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_rs", C.c_int64), ("a_cs", C.c_int64), ("b_rs", C.c_int64), ("b_cs", C.c_int64), ("c_rs", C.c_int64),
        ("dtA", C.c_int32), ("dtB", C.c_int32), ("dtC", C.c_int32),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("batch1", C.c_int32), ("batch2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64), ("sBias1", C.c_int64), ("sBias2", C.c_int64),
        ("act_grad_pre", C.c_void_p), ("dtPre", C.c_int32), ("act_grad_kind", C.c_int32), ("accumulate", C.c_int32),
        ("pre_out", C.c_void_p), ("dtRes", C.c_int32), ("pre_out_kind", C.c_int32),
    ]


_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64


class TnItem(C.Structure):
    """demo_tn_item (include/demo_hip.h): one weight-gradient product of a grouped launch."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("Mo", C.c_int32), ("No", C.c_int32),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("colsum", C.c_void_p)]


# name -> argtypes (restype is int for all but demo_last_error); mirrors include/demo_hip.h
PROTOTYPES = {
    "demo_abi_version": [],
    "demo_gemm": [C.POINTER(GemmDesc), _P],
    "demo_gemm_desc_size": [],
    "demo_gemm_force_tile": [C.c_int],
    "demo_gemm_path": [C.POINTER(GemmDesc)],
    "demo_gemm_split_rows": [C.POINTER(GemmDesc)],
    "demo_gemm_split_plan": [C.POINTER(GemmDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32],
    "demo_gemm_set_split_equal": [C.c_int],
    "demo_gemm_set_rem160": [C.c_int],
    "demo_gemm_set_split": [C.c_int],
    "demo_gemm_set_pp_epilogue": [C.c_int],
    "demo_gemm_set_pp_min_fill": [C.c_int],
    "demo_gemm_set_dephase": [C.c_int],
    "demo_gemm_set_reverse_m": [C.c_int],
    "demo_gemm_set_group_m": [C.c_int],
    "demo_gemm_generic_tile": [C.c_int],
    "demo_gemm_set_colslice": [C.c_int],
    "demo_gemm_set_epi_stage": [C.c_int],
    "demo_gemm_set_split_overlap": [C.c_int],
    "demo_gemm_store_policy": [],
    "demo_cast_f32_bf16": [_P, _P, _L, _P],
    "demo_cast_bf16_f32": [_P, _P, _L, _P],
    "demo_droppath_mask": [_P, _I, _F, _U64, _P, _P],
    "demo_rowscale_add": [_P, _P, _I, _P, _P, _I, _I, _I, _P],
    "demo_rowscale_cast": [_P, _P, _P, _I, _I, _I, _I, _P],
    "demo_transpose": [_P, _P, _I, _I, _I, _P],
    "demo_transpose_batched": [_P, _P, _I, _I, _I, _L, _L, _I, _P],
    "demo_colsum": [_P, _P, _I, _I, _L, _I, _I, _P],
    "demo_im2col": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "demo_embed_tokens": [_P, _I, _P, _P, _P, _P, _I, _F, _I, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "demo_layernorm_fwd": [_P, _P, _P, _F, _P, _I, _P, _P, _I, _I, _P],
    "demo_layernorm_bwd": [_P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P],
    "demo_layernorm_bwd_workspace": [_I, _I],
    "demo_layernorm_bwd_lp": [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "demo_layernorm_add_fwd": [_P, _I, _P, _I, _P, _P, _P, _F, _P, _I, _P, _P, _I, _I, _P],
    "demo_adam_step": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _I, _P, _P],
    "demo_optim_step": [_I, _P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _I, _P, _P, _P, _P],
    "demo_batchnorm_fwd": [_P, _P, _P, _P, _P, _I, _F, _F, _P, _I, _P, _P, _I, _I, _I, _I, _I, _L, _L, _P, _P],
    "demo_attention_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "demo_attention_set_variant": [C.c_int],
    "demo_layernorm_set_lp8": [C.c_int],
    "demo_pife_pool_cat": [_P, _I, _P, _I, _I, _I, _P],
    "demo_set_token0": [_P, _I, _P, _L, _I, _I, _I, _P],
    "demo_get_token0": [_P, _I, _P, _L, _I, _I, _I, _P],
    "demo_dgaf_attn_pool": [_P, _L, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "demo_dgaf_attn_pool_bwd": [_P, _P, _P, _P, _I, _P, _L, _P, _I, _I, _I, _I, _P],
    "demo_dgaf_gate_fuse": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _P],
    "demo_dgaf_gate_fuse_bwd": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _P],
    "demo_dgaf_combine": [_P, _P, _P, _I, _I, _P],
    "demo_dgaf_sum3": [_P, _P, _I, _I, _P],
    "demo_hdm_softmax_pool": [_P, _L, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _U64, _P, _P],
    "demo_hdm_softmax_pool_bwd": [_P, _P, _P, _P, _I, _P, _P, _P, _L, _P, _I, _I, _I, _I, _F, _U64, _P, _P],
    "demo_hdm_vbias_bwd": [_P, _P, _L, _P, _P, _P, _I, _I, _I, _P],
    "demo_atmoe_gate_softmax_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "demo_gemm_tn_bf16": [_P, _P, _P, _I, _I, _I, _L, _L, _L, _F, _I, _P, _P, _L, _P],
    "demo_gemm_tn_workspace": [_I, _I, _I],
    "demo_gemm_tn_bf16_group": [C.POINTER(TnItem), _I, _I, _F, _I, _P, _L, _P],
    "demo_gemm_tn_group_workspace": [C.POINTER(TnItem), _I, _I],
    "demo_attention_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "demo_embed_tokens_bwd": [_P, _P, _I, _F, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P],
    "demo_act_bwd": [_P, _P, _I, _P, _L, _P],
    "demo_act_fwd": [_P, _I, _P, _L, _P],
    "demo_batchnorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _L, _P, _P, _P],
    "demo_token0_grad": [_P, _P, _L, _I, _I, _I, _P],
    "demo_tokens_grad": [_P, _P, _L, _P, _L, _I, _P, _I, _I, _I, _I, _P],
    "demo_hdm_vbias": [_P, _P, _L, _P, _I, _I, _I, _P],
    "demo_atmoe_gate_softmax": [_P, _P, _P, _I, _I, _I, _P],
    "demo_atmoe_eval_supported": [_I, _I],
    "demo_atmoe_eval_part_elems": [_I, _I],
    "demo_atmoe_eval_fwd": [_P] * 14 + [_L, _L, _F] + [_P] * 5 + [_I, _I, _I, _P],
    "demo_xent_label_smooth": [_P, _L, _P, _I, _I, _F, _F, _P, _P, _L, _P, _P],
    "demo_triplet_soft_margin": [_P, _L, _P, _I, _I, _F, _P, _P, _L, _P, _P],
    "demo_triplet_loss": [_P, _L, _P, _I, _I, _F, _F, _P, _P, _L, _P, _P],
    "demo_l2_normalize_rows": [_P, _L, _I, _I, _F, _P],
    "demo_row_sqnorm": [_P, _L, _I, _I, _P, _P],
    "demo_add_outer_sum": [_P, _L, _P, _P, _I, _I, _P],
    "demo_rank_eval": [_P, _L, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
}

_lib = None


def header_symbols():
    """Every function name declared in include/demo_hip.h (used by the CPU export test)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(demo_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (once) and return the ctypes handle; raises DemoHipError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DemoHipError(f"{LIB_PATH} not found: build the HIP extension first (make -C demo2_amd/csrc). "
                           "There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.demo_layernorm_bwd_workspace.restype = C.c_int64
    lib.demo_gemm_tn_workspace.restype = C.c_int64
    lib.demo_gemm_tn_group_workspace.restype = C.c_int64
    lib.demo_atmoe_eval_part_elems.restype = C.c_int64
    lib.demo_last_error.argtypes = []
    lib.demo_last_error.restype = C.c_char_p
    if lib.demo_gemm_desc_size() != C.sizeof(GemmDesc):
        raise DemoHipError("demo_gemm_desc layout mismatch between include/demo_hip.h and demo2_amd/_lib.py")
    # tuning hooks for same-box A/B runs of whole steps (defaults are the measured-best settings)
    if "DEMO_PP_EPILOGUE" in os.environ:
        lib.demo_gemm_set_pp_epilogue(int(os.environ["DEMO_PP_EPILOGUE"]))      # 0: LDS-staged epilogue of the 256^2 NT kernel
    if "DEMO_PP_MIN_FILL" in os.environ:
        lib.demo_gemm_set_pp_min_fill(int(os.environ["DEMO_PP_MIN_FILL"]))      # 85: the pre-register-epilogue kernel choice
    if "DEMO_ATTN_VARIANT" in os.environ:
        lib.demo_attention_set_variant(int(os.environ["DEMO_ATTN_VARIANT"]))    # 0: round-1 attention forward (Q, K, V in LDS)
    if "DEMO_GEMM_SPLIT_EQUAL" in os.environ:
        lib.demo_gemm_set_split_equal(int(os.environ["DEMO_GEMM_SPLIT_EQUAL"]))   # partial-round products as equal 256^2 launches up to this K (0 = never)
    if "DEMO_GEMM_GROUP_M" in os.environ:
        lib.demo_gemm_set_group_m(int(os.environ["DEMO_GEMM_GROUP_M"]))         # 256x256 kernel: row panels per tile-order group
    if "DEMO_GEMM_REVERSE_M" in os.environ:
        lib.demo_gemm_set_reverse_m(int(os.environ["DEMO_GEMM_REVERSE_M"]))     # 256x256 kernel: reverse row-panel walk from this K
    if "DEMO_GEMM_REM160" in os.environ:
        lib.demo_gemm_set_rem160(int(os.environ["DEMO_GEMM_REM160"]))           # 0: row-split remainders always on the 128x128 kernel
    if "DEMO_GENERIC_TILE" in os.environ:
        lib.demo_gemm_generic_tile(int(os.environ["DEMO_GENERIC_TILE"]))        # generic fp32 kernel: 32 / 64 forced (0 = heuristic)
    if "DEMO_GEMM_DEPHASE" in os.environ:
        lib.demo_gemm_set_dephase(int(os.environ["DEMO_GEMM_DEPHASE"]))         # 256x256 kernel: start offsets of the first round
    if "DEMO_GEMM_SPLIT_OVERLAP" in os.environ:
        lib.demo_gemm_set_split_overlap(int(os.environ["DEMO_GEMM_SPLIT_OVERLAP"]))   # 1: split remainders on a side stream beside the main part (experiment)
    if "DEMO_GEMM_EPI_STAGE" in os.environ:
        lib.demo_gemm_set_epi_stage(int(os.environ["DEMO_GEMM_EPI_STAGE"]))     # 0: stores straight from the accumulator layout
    if "DEMO_GEMM_COLSLICE" in os.environ:
        lib.demo_gemm_set_colslice(int(os.environ["DEMO_GEMM_COLSLICE"]))       # 256x256 kernel: weight-panel slices per launch (1 = off)
    if "DEMO_GEMM_SPLIT" in os.environ:
        lib.demo_gemm_set_split(int(os.environ["DEMO_GEMM_SPLIT"]))             # 0: no row split of partial-round products
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().demo_last_error()
        raise DemoHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
