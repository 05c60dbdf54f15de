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
         ("dqkv->dh1", W, 3 * W, "plain"), ("dfc1->dh2", W, 4 * W, "plain"), ("proj fwd", W, W, "bias+res")]
lefts = [int(v) for v in a.left.split(",")]
for name, n, k, epi in cases:
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    res = torch.randn(M, n, device=dev) if "res" in epi else None
    odt = torch.float32 if res is not None else torch.bfloat16

    def run(out, pre):
        ops.gemm(x, w, out=out, bias=bias if "bias" in epi else None, residual=res, act=ACT_QUICKGELU if "act" in epi else 0,
                 pre_out=pre)

    outs = []
    for left in [0] + lefts:
        lib.demo_gemm_set_split(left)
        out = torch.empty(M, n, device=dev, dtype=odt)
        pre = torch.empty(M, n, device=dev, dtype=odt) if "pre" in epi else None
        run(out, pre)
        outs.append((out.float(), None if pre is None else pre.float()))
    for o, p_ in outs[1:]:
        assert (o - outs[0][0]).abs().max().item() <= 2e-2 * outs[0][0].abs().max().item(), name
        if p_ is not None:
            assert (p_ - outs[0][1]).abs().max().item() <= 2e-2 * outs[0][1].abs().max().item(), name
    out = torch.empty(M, n, device=dev, dtype=odt)
    pre = torch.empty(M, n, device=dev, dtype=odt) if "pre" in epi else None
    t = {}
    for rnd in range(3):
        for left in [0] + lefts:
            lib.demo_gemm_set_split(left)
            t.setdefault(left, []).append(timeit(lambda: run(out, pre), a.reps))
    fl = 2.0 * M * n * k
    tiles = -(-M // 256) * (n // 256)
    print(f"{name:10s} N={n:5d} K={k:5d} {epi:13s} tiles {tiles:5d} ({tiles / 256:5.2f} rounds): " +
          "  ".join(f"left<={lf}: {min(v) * 1e3:6.1f} us {fl / min(v) / 1e9:5.0f} TF" for lf, v in t.items()), flush=True)
lib.demo_gemm_set_split(128)
