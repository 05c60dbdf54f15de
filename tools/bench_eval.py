#!/usr/bin/env python3
"""Evaluator timing at RGBNT100 scale (1715 queries x 8575 gallery, 5120-d features): GPU path vs the numpy oracle.
The oracle is used here as the CPU baseline and checker beside the GPU path (like bench.py's cpu_baseline leg), never as the product."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2_amd.utils.metrics import R1_mAP_eval
from oracle import demo_oracle as O
nq, ng, D = 1715, 8575, 5120
rng = np.random.default_rng(0)
pids = rng.integers(0, 50, nq + ng); camids = rng.integers(0, 8, nq + ng)
centers = rng.normal(size=(50, D)).astype(np.float32)
feats = torch.from_numpy(centers[pids] + 2.0 * rng.normal(size=(nq + ng, D)).astype(np.float32))
fd = feats.cuda()
for rep in range(2):
    ev = R1_mAP_eval(nq, 50, True); ev.update((fd, pids, camids, None))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cmc, mAP, *_ = ev.compute()
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"GPU evaluator: {1e3 * (t1 - t0):.1f} ms   mAP {mAP:.6f} R1 {cmc[0]:.6f}")
t0 = time.perf_counter(); cmc_r, map_r = O.r1_map(feats, pids, camids, nq); t1 = time.perf_counter()
print(f"numpy oracle : {1e3 * (t1 - t0):.1f} ms   mAP {map_r:.6f} R1 {cmc_r[0]:.6f}   |dmAP| {abs(mAP - map_r):.2e}")
