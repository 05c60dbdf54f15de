#!/usr/bin/env python3
"""Does a 256^2 tile run faster when fewer CUs are active?  One round of the 256x256 NT kernel with 64 ... 256 tiles (N = 768: three
column tiles per row panel) for the step's reduction lengths: time per launch = time per tile at that number of active CUs.  If the
main loop is bound by the chip-wide L2 -> LDS feed, the per-tile time falls with the number of active CUs and a grid of 194 workgroups
walking three tiles each (582 = 3 x 194 tiles of an N = 768 product at M = 49536) beats two full rounds plus a remainder.
    python tools/bench_cu_fill.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
W = 768


def timeit(fn, reps=20, rounds=5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


lib.demo_gemm_force_tile(2566)
lib.demo_gemm_set_split(0)
g = torch.Generator(device=dev).manual_seed(0)
try:
    for K in (768, 2304, 3072):
        w = (torch.randn(W, K, device=dev, generator=g) * K ** -0.5).bfloat16()
        for panels in (21, 43, 64, 65, 75, 85, 170):          # x 3 column tiles = 63, 129, 192, 195, 225, 255, 510 tiles
            M = panels * 256
            a = torch.randn(M, K, device=dev, generator=g).bfloat16()
            out = torch.empty(M, W, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: ops.gemm(a, w, out=out))
            tiles = panels * 3
            rounds = -(-tiles // 256)
            print(f"K={K:5d} tiles {tiles:4d} ({rounds} round{'s' if rounds > 1 else ' '}): {t:7.1f} us per launch  {t / rounds:7.1f} us per round  "
                  f"{2.0 * M * W * K / t / 1e6:6.0f} TF   {tiles * (K // 64) * 65536 / t / 1e6:6.2f} TB/s L2->LDS", flush=True)
finally:
    lib.demo_gemm_force_tile(0)
    lib.demo_gemm_set_split(128)
