#!/usr/bin/env python3
"""Self-attention kernels at the training step's shape (3 x 128 images x 129 tokens x 12 heads, bf16): forward variants 0 / 1 and the
backward, interleaved rounds in ONE process (guide rule 24), median / min per variant, algorithmic bytes / time.
    python tools/bench_attn.py [--bn 384] [--rounds 30]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demo2_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bn", type=int, default=384)
    ap.add_argument("--L", type=int, default=129)
    ap.add_argument("--H", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=30)
    a = ap.parse_args()
    lib = _lib.load()
    Bn, L, H = a.bn, a.L, a.H
    D = H * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = (torch.randn(Bn * L, 3 * D, device="cuda", generator=g) * 1.2).bfloat16()
    dout = torch.randn(Bn * L, D, device="cuda", generator=g).bfloat16()
    out, lse = ops.attention_fwd(qkv, Bn, L, H, save_lse=True)
    fwd_bytes = Bn * L * (3 * D + D) * 2 + Bn * H * L * 4
    bwd_bytes = Bn * L * (3 * D + D + D + 3 * D) * 2 + Bn * H * L * 4

    def fwd(v):
        lib.demo_attention_set_variant(v)
        ops.attention_fwd(qkv, Bn, L, H, save_lse=True)

    def bwd(v):
        if hasattr(lib, "demo_attention_bwd_set_variant"):
            lib.demo_attention_bwd_set_variant(v)
        ops.attention_bwd(qkv, out, dout, lse, Bn, L, H)

    arms = [("fwd v0", lambda: fwd(0), fwd_bytes), ("fwd v1", lambda: fwd(1), fwd_bytes), ("bwd v0", lambda: bwd(0), bwd_bytes)]
    if hasattr(lib, "demo_attention_bwd_set_variant"):
        arms.append(("bwd v1", lambda: bwd(1), bwd_bytes))
    # correctness of the variants against each other first
    lib.demo_attention_set_variant(0)
    o0, l0 = ops.attention_fwd(qkv, Bn, L, H, save_lse=True)
    lib.demo_attention_set_variant(1)
    o1, l1 = ops.attention_fwd(qkv, Bn, L, H, save_lse=True)
    print("fwd v1 vs v0: max |dout|", float((o0.float() - o1.float()).abs().max()), "max |dlse|", float((l0 - l1).abs().max()))
    times = {n: [] for n, _, _ in arms}
    for _ in range(3):
        for n, f, _ in arms:
            f()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for n, f, _ in arms:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                f()
            e1.record()
            e1.synchronize()
            times[n].append(e0.elapsed_time(e1) * 1e3 / 4)
    lib.demo_attention_set_variant(1)
    for n, _, nb in arms:
        t = times[n]
        med, mn = statistics.median(t), min(t)
        print(f"{n}: median {med:7.1f} us  min {mn:7.1f} us   {nb / 1e6:6.1f} MB -> {nb / med / 1e6:5.2f} TB/s (median)")


if __name__ == "__main__":
    main()
