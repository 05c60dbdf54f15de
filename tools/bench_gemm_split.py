#!/usr/bin/env python3
"""Row split of the 256^2 NT launches (csrc/gemm_bf16.hip: demo_gemm_bf16_split_rows): correctness of the two-launch form
against the one-launch form, then timing with the split off / on for the shapes and epilogues of the training step.
    python tools/bench_gemm_split.py [--batch 128] [--reps 10] [--left 64,128,200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402
from demo2_amd.ops import ACT_QUICKGELU  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--left", default="128,250")
a = ap.parse_args()
M, W = 3 * a.batch * 129, 768
dev = "cuda"
lib = _lib.load()
torch.manual_seed(0)


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# name, N, K, epilogue
cases = [("qkv fwd", 3 * W, W, "bias"), ("fc1 fwd", 4 * W, W, "bias+act+pre"), ("fc2 fwd", W, 4 * W, "bias+res"),
         ("dqkv->dh1", W, 3 * W, "plain"), ("dfc1->dh2", W, 4 * W, "plain"), ("proj fwd", W, W, "bias+res"),
         ("dfc2->dpre", 4 * W, W, "actgrad"), ("dproj->da", W, W, "plain-force"), ("clipproj", 512, W, "plain-force")]
lefts = [int(v) for v in a.left.split(",")]
# variants: (label, register epilogue?, split leftover, forced tile)
variants = [("lds", 0, 0, 0), ("lds+split", 0, 128, 0), ("reg", 1, 0, 0), ("reg+split", 1, 128, 0)]
for name, n, k, epi in cases:
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    res = torch.randn(M, n, device=dev) if "res" in epi else None
    odt = torch.float32 if res is not None else torch.bfloat16

    gpre = torch.randn(M, n, device=dev).bfloat16() if epi == "actgrad" else None

    def run(out, pre):
        ops.gemm(x, w, out=out, bias=bias if "bias" in epi else None, residual=res, act=ACT_QUICKGELU if "act+" in epi else 0,
                 pre_out=pre, act_grad_pre=gpre, act_grad_kind=ACT_QUICKGELU if gpre is not None else 0)

    vs = list(variants)
    if epi == "plain-force":                              # short-K products the heuristic gives to the 128^2 kernel
        vs = [("128^2", 1, 0, 1282), ("256^2 reg", 1, 0, 2566)]
    if epi == "actgrad":                                  # the dispatcher's default for this epilogue is the 128^2 kernel
        vs = [("128^2 (default)", 1, 0, 0), ("256^2 lds", 0, 0, 2566), ("256^2 reg", 1, 0, 2566)]

    def setup(v):
        lib.demo_gemm_set_pp_epilogue(v[1])
        lib.demo_gemm_set_split(v[2])
        lib.demo_gemm_force_tile(v[3])

    outs = []
    for v in vs:
        setup(v)
        out = torch.empty(M, n, device=dev, dtype=odt)
        pre = torch.empty(M, n, device=dev, dtype=odt) if "pre" in epi else None
        run(out, pre)
        outs.append((out.float(), None if pre is None else pre.float()))
    for (o, p_), v in zip(outs[1:], vs[1:]):
        err = (o - outs[0][0]).abs().max().item() / outs[0][0].abs().max().item()
        assert err <= 2e-2, (name, v[0], err)
        if p_ is not None:
            assert (p_ - outs[0][1]).abs().max().item() <= 2e-2 * outs[0][1].abs().max().item(), (name, v[0])
    out = torch.empty(M, n, device=dev, dtype=odt)
    pre = torch.empty(M, n, device=dev, dtype=odt) if "pre" in epi else None
    t = {}
    for rnd in range(3):
        for v in vs:
            setup(v)
            t.setdefault(v[0], []).append(timeit(lambda: run(out, pre), a.reps))
    fl = 2.0 * M * n * k
    tiles = -(-M // 256) * (n // 256)
    print(f"{name:10s} N={n:5d} K={k:5d} {epi:13s} tiles {tiles:5d} ({tiles / 256:5.2f} rounds): " +
          "  ".join(f"{lf}: {min(v) * 1e3:6.1f} us {fl / min(v) / 1e9:5.0f} TF" for lf, v in t.items()), flush=True)
lib.demo_gemm_force_tile(0)
lib.demo_gemm_set_pp_epilogue(1)
lib.demo_gemm_set_split(128)
