#!/usr/bin/env python3
"""Headline benchmark: tri-modal images/sec of the PIFE -> HDM -> ATMoE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode train|fwd] [--batch B]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL).  A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
  mode train (default when the backward kernels are present): forward + backward (+ RCCL gradient all-reduce for N>1)
       of the RGBNT201 DeMo config at B=128 triplets per GPU  (BASELINE.json configs[2] / configs[3]);
  mode fwd : eval forward at B=64 triplets per GPU, bf16                      (BASELINE.json configs[1]).
Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline"     : the dominant kernel (bf16 MFMA GEMM) timed live with HIP events on the launch stream,
  "cpu_baseline" : the CPU oracle (oracle/demo_oracle.py, kind "port") timed on this host for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
B_TRAFFIC_VALID = [False]        # the committed PMC traffic figures describe the default workload only (train step, B=128)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["auto", "train", "fwd"], default="auto")
    ap.add_argument("--batch", type=int, default=0, help="triplets per GPU (default: 128 train / 64 fwd)")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured hipGraph (single GPU): the training step measured equal to eager launches; the eval forward "
                         "(--mode fwd) is host-bound at small batches and gains")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--fusion", choices=["atmoe", "dgaf"], default="atmoe",
                    help="atmoe: the headline PIFE -> HDM -> ATMoE model; dgaf: the DeMo class's DGAF-V3 head (SURVEY 8(f) row 3, a side "
                         "measurement, not BASELINE.json's metric)")
    ap.add_argument("--eager-baseline", action="store_true",
                    help="also time the torch restatement of the path with PyTorch-ROCm eager on this GPU (bf16 autocast); baseline only")
    ap.add_argument("--config", choices=["RGBNT201", "RGBNT100", "vit"], default="RGBNT201",
                    help="RGBNT201: the headline workload (configs/RGBNT201/DeMo.yml, CLIP ViT-B/16, 256x128, HEAD 4, DIRECT 1).  Side lines, NOT "
                         "BASELINE.json's metric: RGBNT100 = configs/RGBNT100/DeMo.yml (128x256, HEAD 8, DIRECT 0, 50 classes, 8 cameras; "
                         "BASELINE configs[4]); vit = the ImageNet ViT-B/16 trunk (C = 768, HEAD 12; SURVEY 8 config V)")
    ap.add_argument("--host-input", action="store_true",
                    help="train mode: copy each batch from pinned host memory (double-buffered side stream); PCIe-inclusive rate")
    return ap.parse_args()


def have_backward():
    try:
        from demo2_amd.modeling import hip_backward  # noqa: F401
        return True
    except ImportError:
        return False


def gemm_roofline(model, step_fn):
    """Instrumented extra (eager) step: every demo_gemm launch that takes an MFMA kernel is bracketed by HIP events on the
    launch stream (torch's current stream IS the stream handed to the C ABI).  The roofline object describes the DOMINANT
    kernel -- the one with the largest summed duration, gemm_bf16_nt_pp_kernel (256x256 ping-pong) on this workload --
    achieved = its algorithmic FLOPs (2*M*N*K per launch) / its summed launch durations; the other MFMA NT kernel is
    reported beside it."""
    from demo2_amd import ops, _lib
    import ctypes as C
    lib = _lib.load()
    recs = {1: [], 2: []}
    orig = lib.demo_gemm

    def timed(d, path):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = orig(C.byref(d), torch.cuda.current_stream().cuda_stream)
        e1.record()
        nb = d.batch1 * d.batch2
        szc = 4 if d.dtC == _lib.F32 else 2
        sz_pre_out = 0 if not d.pre_out else (1 if d.pre_out_kind == 2 else szc)                  # kind 2: DEMO_D8 codes, one byte per element
        sz_pre = 0 if not d.act_grad_pre else {_lib.F32: 4, _lib.BF16: 2, _lib.D8: 1}[d.dtPre]
        nbytes = nb * (2 * d.M * d.K + 2 * d.N * d.K + d.M * d.N * (szc + sz_pre_out)
                       + (d.M * d.N * (2 if d.dtRes == _lib.BF16 else 4) if d.residual else 0) + (d.M * d.N * 4 if d.accumulate else 0)
                       + d.M * d.N * sz_pre)
        epi = "+".join(t for t, on in (("bias", d.bias), ("act", d.act), ("save_pre", d.pre_out and d.pre_out_kind == 0), ("save_deriv", d.pre_out and d.pre_out_kind == 1),
                                       ("save_deriv_d8", d.pre_out and d.pre_out_kind == 2),
                                       ("bf16_residual" if d.dtRes == _lib.BF16 else "fp32_residual", d.residual),
                                       ("act_grad_d8" if d.dtPre == _lib.D8 else "act_grad", d.act_grad_pre), ("accumulate", d.accumulate)) if on) or "plain"
        fam = 1 if (d.pre_out or d.act_grad_pre) else 0        # the 256^2 kernel's epilogue family = its template instance (csrc/gemm_bf16.hip, pp_epi)
        recs[path].append((e0, e1, 2.0 * d.M * d.N * d.K * nb, nbytes, (d.M, d.N, d.K, epi), fam))
        return rc

    class Wrapper:
        def __call__(self, dref, stream):
            d = dref._obj
            path = lib.demo_gemm_path(C.byref(d))
            if path not in recs:
                return orig(dref, stream)
            rows, cfgs = (C.c_int32 * 4)(), (C.c_int32 * 4)()
            nparts = lib.demo_gemm_split_plan(C.byref(d), rows, cfgs, 4)
            if nparts <= 1:
                return timed(d, path)
            # demo_gemm issues this descriptor as several launches over disjoint row ranges (main + remainder on the 128^2 kernel, or
            # equal parts on the 256^2 kernel): issue the same launches from here so that each one gets its own event pair
            szc, szp = (4 if d.dtC == _lib.F32 else 2), {_lib.F32: 4, _lib.BF16: 2, _lib.D8: 1}[d.dtPre] if d.act_grad_pre else 0
            rc, m0 = 0, 0
            try:
                for i in range(nparts):
                    di = _lib.GemmDesc.from_buffer_copy(d)
                    di.M = rows[i]
                    di.A = d.A + m0 * d.a_rs * 2
                    di.C = d.C + m0 * d.c_rs * szc
                    if d.residual:
                        di.residual = d.residual + m0 * d.c_rs * (2 if d.dtRes == _lib.BF16 else 4)
                    if d.act_grad_pre:
                        di.act_grad_pre = d.act_grad_pre + m0 * d.c_rs * szp
                    if d.pre_out:
                        di.pre_out = d.pre_out + m0 * d.c_rs * (1 if d.pre_out_kind == 2 else szc)
                    m0 += rows[i]
                    lib.demo_gemm_force_tile(cfgs[i])
                    rc = rc or timed(di, 2 if cfgs[i] == 2566 else 1)
            finally:
                lib.demo_gemm_force_tile(0)
            return rc

    lib.demo_gemm = Wrapper()
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        lib.demo_gemm = orig
    # the 256^2 kernel is ONE template compiled once per epilogue family (<true, 0>: bias / activation / residual; <true, 1>: the MLP's second
    # output or act' operand) so that neither instance carries the other's variants: rocprofv3 lists the two instances as two rows,
    # `instances` below gives each one's launches and average duration, the roofline figures cover both
    names = {1: "gemm_bf16_nt_kernel<128,128,...> / <160,128,...,3> (row-split remainders, small products)", 2: "gemm_bf16_nt_pp_kernel<true, *>"}
    stats = {}
    for path, rs in recs.items():
        if rs:
            ms = sum(r[0].elapsed_time(r[1]) for r in rs)
            fl = sum(r[2] for r in rs)
            shapes = {}
            for r in rs:
                e = shapes.setdefault(r[4], [0, 0.0, 0.0])
                e[0] += 1
                e[1] += r[0].elapsed_time(r[1])
                e[2] += r[2]
            stats[path] = {"kernel": names[path], "launches_per_step": len(rs), "avg_launch_us": round(ms * 1e3 / len(rs), 1),
                           "ms_per_step": round(ms, 3), "algorithmic_gflop_per_step": round(fl / 1e9, 1),
                           "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 1),
                           "algorithmic_mb_per_launch": round(sum(r[3] for r in rs) / len(rs) / 1e6, 1),
                           "instances": ({f"gemm_bf16_nt_pp_kernel<true, {f}>": {
                               "launches_per_step": sum(1 for r in rs if r[5] == f),
                               "avg_launch_us": round(sum(r[0].elapsed_time(r[1]) for r in rs if r[5] == f) * 1e3 / max(1, sum(1 for r in rs if r[5] == f)), 1),
                               "tflops": round(sum(r[2] for r in rs if r[5] == f) / max(1e-9, sum(r[0].elapsed_time(r[1]) for r in rs if r[5] == f) * 1e-3) / 1e12, 1)}
                               for f in (0, 1) if any(r[5] == f for r in rs)} if path == 2 else None),
                           # the same launches by product shape and epilogue (rows M of a split product = the part this kernel computes)
                           "by_shape": [{"M": k[0], "N": k[1], "K": k[2], "epilogue": k[3], "launches_per_step": v[0],
                                         "avg_launch_us": round(v[1] * 1e3 / v[0], 1), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                                        for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:10]]}
    if not stats:
        return None
    dom = max(stats, key=lambda k: stats[k]["ms_per_step"])
    s = stats[dom]
    # HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE doubled for gfx950 + WRITE_SIZE, collected
    # with rocprofv3 in separate runs of this same command and committed under profiles/): not measurable in-process.  The newest
    # profiles/rNN_hbm_traffic_pmc.json is used; `stale` says the kernel sources have changed since it was taken.
    traffic = None
    try:
        import glob
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pmc_traffic import csrc_sha16
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.json")))[-1]
        with open(path) as f:
            doc = json.load(f)
        ks = doc["kernels"]
        inst = [ks[k] for k in ks if k.startswith(s["kernel"].split("<")[0] + "<true")]     # every instance of the template
        t = None
        if inst:
            n = sum(i["launches"] for i in inst)
            t = {"hbm_mb_per_launch": round(sum(i["hbm_mb_per_launch"] * i["launches"] for i in inst) / n, 1)}
        if t and B_TRAFFIC_VALID[0]:
            traffic = {"hbm_mb_per_launch": t["hbm_mb_per_launch"], "algorithmic_mb_per_launch": s["algorithmic_mb_per_launch"],
                       "stale": doc.get("csrc_sha16") != csrc_sha16(),
                       "source": "profiles/" + os.path.basename(path) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, B=128 train step)"}
            for extra in ("fetch_served_by", "note_l2"):
                if extra in t:
                    traffic[extra] = t[extra]
    except (OSError, ValueError, KeyError, IndexError, ImportError):
        pass
    out = {"bound": "mfma", "kernel": s["kernel"], "achieved": s["achieved_tflops"], "peak": MFMA_BF16_PEAK_TFLOPS,
           "unit": "TFLOP/s", "frac": round(s["achieved_tflops"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
           "launches_per_step": s["launches_per_step"], "avg_launch_us": s["avg_launch_us"],
           "gemm_ms_per_step": s["ms_per_step"], "algorithmic_gflop_per_step": s["algorithmic_gflop_per_step"],
           "algorithmic_mb_per_launch": s["algorithmic_mb_per_launch"], "instances": s.get("instances"), "by_shape": s["by_shape"]}
    other = [v for k, v in stats.items() if k != dom]
    if other:
        out["other_mfma_nt_kernel"] = other[0]
    return out


def cpu_baseline(mode, seconds, fusion="atmoe", config="RGBNT201"):
    """The CPU restatement (oracle) on this host's cores, bounded sample of the same workload (B=2 batches)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import demo_oracle as O
    from util import synth_batch
    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    spec = O.Spec() if fusion == "atmoe" else O.Spec(hdm=False, atm=False, use_dgaf=True)
    h, w, cams = 256, 128, 4
    if config == "RGBNT100":
        spec = O.Spec(img_h=128, img_w=256, moe_head=8, direct=0, num_classes=50, camera_num=8)
        h, w, cams = 128, 256, 8
    elif config == "vit":
        spec = O.Spec(arch="vit", width=768, feat_dim=768, moe_head=12, sie_coe=3.0)
    sd = O.init_state_dict(spec, seed=1)
    x, cam = synth_batch(2, h, w, cams)
    target = torch.tensor([0, 0])
    if mode == "train":
        sd = {k: v.requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}

    def one():
        if mode == "train":
            out = O.model_forward(sd, spec, x, cam, training=True, updates={})
            O.train_loss(out, target).backward()
        else:
            with torch.no_grad():
                O.model_forward(sd, spec, x, cam)
    one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 2:
            break
    return {"value": round(2 * n / dt, 2), "unit": "tri-modal images/s (1 image = one RGB+NIR+TIR triplet)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} {'fwd+bwd' if mode == 'train' else 'eval forward'} passes of the fp32 CPU oracle, {config} " + ("DeMo" if fusion == "atmoe" else "DGAF-V3") + f" config, B=2 "
                      f"({dt:.1f} s)"}


def eager_gpu_baseline(mode, B, dev, steps=5):
    """Optional second baseline (--eager-baseline; part of the baseline leg, never the product): the same restatement run by
    PyTorch-ROCm EAGER on this GPU under bf16 autocast (rocBLAS / MIOpen kernels, fused scaled_dot_product_attention in place of
    the oracle's explicit softmax attention, as the reference's nn.MultiheadAttention would dispatch) -- the closest stand-in for
    'the reference itself on an MI355X', which cannot travel to the box.  Same batch size as the measured workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import demo_oracle as O
    from util import synth_batch
    import torch.nn.functional as F
    spec = O.Spec()
    sd = {k: v.to(dev) for k, v in O.init_state_dict(spec, seed=1).items()}
    if mode == "train":
        sd = {k: v.requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    x, cam = synth_batch(B, 256, 128, 4, device=dev)
    target = (torch.arange(B, device=dev) // 4) % 171
    orig = O.mha_self

    def mha_sdpa(xx, in_w, in_b, out_w, out_b, heads):
        Bq, L, D = xx.shape
        qkv = O.linear(xx, in_w, in_b).reshape(Bq, L, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(Bq, L, D)
        return O.linear(o, out_w, out_b)

    O.mha_self = mha_sdpa
    try:
        def one():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if mode == "train":
                    out = O.model_forward(sd, spec, x, cam, training=True, updates={})
                    loss = O.train_loss(out, target)
                else:
                    with torch.no_grad():
                        O.model_forward(sd, spec, x, cam)
            if mode == "train":
                loss.backward()
                for v in sd.values():
                    v.grad = None
        for _ in range(2):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        O.mha_self = orig
    return {"value": round(B / dt, 1), "unit": "tri-modal images/s (1 image = one RGB+NIR+TIR triplet)", "ms_per_step": round(dt * 1e3, 2), "kind": "port, PyTorch-ROCm eager on the GPU",
            "sample": f"{steps} {'fwd+bwd (no optimizer step)' if mode == 'train' else 'eval forward'} passes at B={B}, bf16 autocast, SDPA attention"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # rehearsal of the N > 1 control flow on a one-GPU box: DEMO_BENCH_REHEARSAL=1 puts every rank on cuda:0 and reduces
    # over gloo (functional check only; the numbers mean nothing)
    rehearsal = os.environ.get("DEMO_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    mode = a.mode if a.mode != "auto" else ("train" if have_backward() else "fwd")
    B = a.batch or (128 if mode == "train" else 64)
    B_TRAFFIC_VALID[0] = mode == "train" and B == 128 and a.dtype == "bf16" and a.fusion == "atmoe" and a.config == "RGBNT201"

    # the measured path imports nothing from oracle/ (or tests/): configuration, weights (the modules' own seeded random
    # initialisation -- there are no pretrained files offline) and the synthetic batch are made here
    from demo2_amd.config import get_defaults
    from demo2_amd.modeling import make_model
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "demo2_amd", "configs", "RGBNT100" if a.config == "RGBNT100" else "RGBNT201", "DeMo.yml"))
    n_cls, n_cam = (50, 8) if a.config == "RGBNT100" else (171, 4)       # dataset statistics: SURVEY.md section 8, config symbols
    if a.config == "vit":                                                  # config V: config/defaults.py:86,89
        cfg.MODEL.TRANSFORMER_TYPE, cfg.MODEL.HEAD, cfg.MODEL.SIE_COE = "vit_base_patch16_224", 12, 3.0
    if a.fusion == "dgaf":
        cfg.MODEL.HDM, cfg.MODEL.ATM, cfg.MODEL.USE_DGAF = False, False, True
    img_h, img_w = cfg.INPUT.SIZE_TRAIN
    per_id = int(cfg.DATALOADER.NUM_INSTANCE)
    torch.manual_seed(1234)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # make_model prints the reference's banner: stdout carries ONLY the JSON line
        model = make_model(cfg, n_cls, n_cam, 0)
    model = model.to(dev).set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    gen = torch.Generator().manual_seed(1234 + rank)                      # SOLVER.SEED (config/defaults.py:182)
    x = {k: torch.randn(B, 3, img_h, img_w, generator=gen).to(dev) for k in ("RGB", "NI", "TI")}
    cam = torch.randint(0, n_cam, (B,), generator=gen).to(dev)
    # PK-structured labels: NUM_INSTANCE per identity (configs/*/DeMo.yml)
    target = (torch.arange(B, device=dev) // per_id + rank * max(1, B // per_id)) % n_cls

    if mode == "train":
        from demo2_amd.modeling.hip_backward import TrainStep
        # --graph (single GPU): the step is captured into a hipGraph after two eager steps (needs --warmup >= 3 to stay out of
        # the timed region).  Measured on MI355X: 48.86 ms replayed vs 48.85 ms eager -- the ~3 ms of inter-kernel gaps per
        # step are dispatch latency between dependent kernels, not host launch cost -- so eager launches are the default.
        stepper = TrainStep(model, cfg, world, graph=(world == 1 and a.graph))
        model.train()
        if world > 1:                          # DDP's start-up parameter broadcast (ranks are seeded alike, so a no-op in value)
            from demo2_amd.dist import broadcast_parameters
            broadcast_parameters(model)
            model.refresh_shadow()

        if a.host_input:
            # PCIe-inclusive variant (NOT the contract's `value`): the batch lives in pinned host memory, as train_collate_fn
            # would leave it, and is copied to one of two device buffers on a side stream while the previous step computes
            host = {k: v.cpu().pin_memory() for k, v in x.items()}
            bufs = [{k: torch.empty_like(v) for k, v in x.items()} for _ in range(2)]
            copy_stream, ready, turn = torch.cuda.Stream(), [torch.cuda.Event(), torch.cuda.Event()], [0]

            def upload(i):
                copy_stream.wait_stream(torch.cuda.current_stream())      # the buffer's previous consumer has been queued
                with torch.cuda.stream(copy_stream):
                    for k in host:
                        bufs[i][k].copy_(host[k], non_blocking=True)
                    ready[i].record()
            upload(0)

            def step():
                i = turn[0]
                torch.cuda.current_stream().wait_event(ready[i])
                upload(1 - i)                                             # next batch flies under this step
                stepper.step(bufs[i], target, cam)
                turn[0] = 1 - i
        else:
            def step():
                stepper.step(x, target, cam)

        def eager_step():                      # the instrumented pass must see every launch: no graph replay
            stepper._run(x, target, cam, stepper.hyper if stepper.use_graph else None)
    else:
        model.eval()

        def eager_step():
            with torch.no_grad():
                model(x, cam_label=cam, return_pattern=3)
        step = eager_step
        if a.graph and world == 1:             # small-batch serving: replay the forward from a captured hipGraph (host-bound at B = 1)
            from demo2_amd.modeling.eval_graph import EvalGraph
            replay = EvalGraph(model, B, img_h, img_w, return_pattern=3)

            def step():
                replay(x, cam_label=cam)

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    log(f"mode={mode} B={B} world={world}: model on {dev}, starting {a.warmup} warmup steps")
    for _ in range(a.warmup):
        step()
    fence()
    log("warmup done, timing")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    log(f"timed region: {dt:.3f} s for {a.steps} steps")
    roof = cpu = cpu_fwd = eager = None
    if rank == 0:
        # the instrumented extra step runs on rank 0 ONLY: it must not issue collectives the other ranks never join
        saved_sync = getattr(model, "grad_sync", None)
        model.grad_sync = False                # explicit opt-out (None would make the forward install a GradSync in a multi-rank job)
        try:
            roof = gemm_roofline(model, eager_step)
        finally:
            model.grad_sync = saved_sync
        log("roofline pass done; cpu baseline next")
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(mode, a.cpu_seconds, a.fusion, a.config)
            if mode == "train":
                # BASELINE.json configs[0]: the reference's own CPU-runnable case is the FORWARD on B=2 batches (4 triplets and more)
                cpu_fwd = cpu_baseline("fwd", max(4.0, a.cpu_seconds / 2), a.fusion, a.config)
        if world == 1 and a.eager_baseline and a.fusion == "atmoe":
            eager = eager_gpu_baseline(mode, B, dev)
    if rank == 0:
        ms = dt / a.steps * 1e3
        val = B * world * a.steps / dt
        # config A / B 69.99 G, config V 71.73 G per triplet forward (HDM counted UNFOLDED, as the survey does);
        # DGAF head: trunk 68.34 G + 0.05 G folded fusion.  `model_tflops_folded` counts the algorithm the kernels execute (HDM in its
        # folded single-query form): 68.39 G (A / B), 68.14 G (V) per triplet forward.
        unfolded = {"RGBNT201": 69.99e9, "RGBNT100": 69.99e9, "vit": 71.73e9}[a.config] if a.fusion == "atmoe" else 68.39e9
        folded = {"RGBNT201": 68.39e9, "RGBNT100": 68.39e9, "vit": 68.14e9}[a.config] if a.fusion == "atmoe" else 68.39e9
        flop_per_triplet = unfolded * (3 if mode == "train" else 1)
        flop_folded = folded * (3 if mode == "train" else 1)
        line = {
            "metric": "tri-modal images/sec " + ("fwd+bwd" if mode == "train" else "fwd") + " (ViT-B, RGBNT201 shape)",
            "value": round(val, 1), "unit": "tri-modal images/s (1 image = one RGB+NIR+TIR triplet)", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ((({"RGBNT201": "RGBNT201 DeMo.yml (CLIP ViT-B/16 x3 + PIFE + HDM + ATMoE), ",
                                       "RGBNT100": "RGBNT100 DeMo.yml (CLIP ViT-B/16 x3 + PIFE + HDM + ATMoE, HEAD 8, DIRECT 0; side line), ",
                                       "vit": "RGBNT201 geometry with the ImageNet ViT-B/16 trunk (C 768, HEAD 12; side line), "}[a.config])
                                     if a.fusion == "atmoe" else
                                     "RGBNT201 geometry, DeMo class with MODEL.USE_DGAF (CLIP ViT-B/16 x3 + DGAF-V3), ")
                                    + ("train step fwd+bwd, " if mode == "train" else "eval forward, ")
                                    + f"3x({B},3,{img_h},{img_w}) per GPU, random-init weights"),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}", "mode": mode,
                       "launch": "hipGraph replay" if ((mode == "train" and stepper.use_graph) or (mode == "fwd" and a.graph and world == 1)) else "eager",
                       "input": "pinned host memory, copied every step (PCIe-inclusive)" if (mode == "train" and a.host_input)
                       else "resident in HBM"},
            "model_tflops": round(val * flop_per_triplet / 1e12, 1),
            "model_tflops_folded": round(val * flop_folded / 1e12, 1),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if cpu_fwd is not None:
            line["cpu_baseline_forward"] = cpu_fwd
        if eager is not None:
            line["eager_gpu_baseline"] = eager
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
