"""Golden-vector generator (TEST INFRASTRUCTURE; runs only where /root/reference exists).

Imports the reference's own torch-only files *by path* on CPU (never the package
`modeling/__init__.py`, which needs fvcore/timm/torchvision that are absent here),
seeds torch, runs them on small synthetic inputs and stores inputs, state-dicts and
outputs as fp32 `.npz` fixtures under tests/golden/.  Nothing from the reference is
copied: fixtures are data (tensors), and the reference never travels to the GPU box.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = os.environ.get("DEMO_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True


def _load(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace of the reference modules on the hot path."""
    for pkg in ("modeling", "modeling.backbones", "modeling.clip", "modeling.moe"):
        if pkg not in sys.modules:                       # namespace shims: package __init__ not executed
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    ns = SimpleNamespace()
    ns.attnmoe = _load("modeling.moe.AttnMOE", "modeling/moe/AttnMOE.py")
    ns.vit = _load("modeling.backbones.vit_pytorch", "modeling/backbones/vit_pytorch.py")
    ns.clip = _load("modeling.clip.model", "modeling/clip/model.py")
    ns.triplet = _load("ref_triplet_loss", "layers/triplet_loss.py")
    ns.softmax = _load("ref_softmax_loss", "layers/softmax_loss.py")
    return ns


def load_reference_metrics():
    """utils/metrics.py with plotting-only imports stubbed (they are never called by eval_func)."""
    for name in ("seaborn", "matplotlib", "matplotlib.patches", "matplotlib.pyplot", "PIL", "PIL.Image"):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if "utils" not in sys.modules or not hasattr(sys.modules["utils"], "__path__"):
        u = types.ModuleType("utils")
        u.__path__ = []
        sys.modules["utils"] = u
    rr = types.ModuleType("utils.reranking")
    rr.re_ranking = None
    sys.modules["utils.reranking"] = rr
    return _load("ref_metrics", "utils/metrics.py")


def _np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}::{kk}"] = np.asarray(vv)
        else:
            flat[k] = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def _randomise_bn(module, gen):
    """Make BatchNorm running stats / affine non-trivial so eval-mode parity means something."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
            m.running_var.copy_((1 + torch.randn(m.num_features, generator=gen) * 0.1).abs())
            m.weight.data.copy_(1 + torch.randn(m.num_features, generator=gen) * 0.05)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.05)


def gen_fusion(ref, C=128, H=4, n_tok=8, tag="fusion_c128_h4", with_grads=False):
    """GeneralFusion (HDM + ATMoE): eval at B in {1,2,4}; train at B=4 (dropout off so
    that the output is a deterministic function of the inputs) incl. BN running-stat updates;
    HDM-only mode; intermediate HDM features and gate."""
    torch.manual_seed(7)
    gen = torch.Generator().manual_seed(11)
    cfg = SimpleNamespace(MODEL=SimpleNamespace(HDM=True, ATM=True))
    gf = ref.attnmoe.GeneralFusion(feat_dim=C, num_experts=7, head=H, reg_weight=0, dropout=0.0, cfg=cfg)
    with torch.no_grad():
        _randomise_bn(gf, gen)
        for n, p in gf.named_parameters():
            if n.endswith("in_proj_bias") or n.endswith("out_proj.bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)     # defaults are zero
    arrays = {"sd": _np(gf.state_dict())}
    gf.eval()
    for B in (1, 2, 4):
        ins = [torch.randn(B, n_tok, C, generator=gen) for _ in range(3)] + [torch.randn(B, C, generator=gen) for _ in range(3)]
        with torch.no_grad():
            feats = gf.forward_HDM(*ins)
            out = gf(*ins)
            f2 = [f.unsqueeze(0) if f.dim() == 1 else f for f in feats]
            gate = gf.moe.gating_network(torch.cat(f2, -1), torch.stack(f2, 1))
        arrays[f"eval_b{B}"] = {**{f"in{i}": t.numpy() for i, t in enumerate(ins)},
                                **{f"hdm{i}": t.numpy() for i, t in enumerate(f2)},
                                "gate": gate.numpy(), "out": out.numpy()}
    # HDM-only (ATM False) path, eval
    gf.ATM = False
    with torch.no_grad():
        arrays["eval_b4"]["out_hdm_only"] = gf(*ins).numpy()
    gf.ATM = True
    # train mode, B=4, dropout 0
    gf.train()
    ins = [torch.randn(4, n_tok, C, generator=gen) for _ in range(3)] + [torch.randn(4, C, generator=gen) for _ in range(3)]
    for t in ins:
        t.requires_grad_(True)
    out, loss = gf(*ins)
    w = torch.randn(out.shape, generator=gen)
    (out * w).sum().backward()
    arrays["train_b4"] = {**{f"in{i}": t.detach().numpy() for i, t in enumerate(ins)},
                          "out": out.detach().numpy(), "loss": np.asarray(float(loss)), "w": w.numpy(),
                          **{f"gin{i}": t.grad.numpy() for i, t in enumerate(ins)}}
    arrays["sd_after_train"] = {k: v for k, v in _np(gf.state_dict()).items() if "running" in k or "num_batches" in k}
    if with_grads:
        arrays["grads"] = {n: p.grad.numpy() for n, p in gf.named_parameters() if p.grad is not None}
    _save(tag, **arrays)


def gen_dgaf(tag="dgaf_v3_c64_h4"):
    """DualGatedAdaptiveFusionV3 (modeling/dual_gated_fusion.py:558-745), output_dim = 3C: forward at B in {1,3} and
    a backward (gradients w.r.t. inputs and every parameter) under a fixed random cotangent."""
    dg = _load("ref_dual_gated_fusion", "modeling/dual_gated_fusion.py")
    torch.manual_seed(41)
    gen = torch.Generator().manual_seed(43)
    C, H, K = 64, 4, 9
    m = dg.DualGatedAdaptiveFusionV3(feat_dim=C, output_dim=3 * C, tau=0.7, init_alpha=0.3, num_heads=H)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)     # defaults are zero
            if ("norm" in n or n in ("gate_net.1.weight", "modal_enhance.1.weight")) and n.endswith("weight"):
                p.copy_(1 + torch.randn(p.shape, generator=gen) * 0.05)
            if n.endswith("_query"):
                p.copy_(torch.randn(p.shape, generator=gen))            # the default scale C^-0.5 leaves the pooling nearly uniform
    arrays = {"sd": _np(m.state_dict()), "meta": np.asarray([C, H, K, 0.7])}
    m.eval()
    for B in (1, 3):
        ins = [torch.randn(B, K, C, generator=gen) for _ in range(3)]
        with torch.no_grad():
            out = m(*ins)
        arrays[f"eval_b{B}"] = {**{f"in{i}": t.numpy() for i, t in enumerate(ins)}, "out": out.numpy()}
    m.train()
    ins = [torch.randn(4, K, C, generator=gen).requires_grad_(True) for _ in range(3)]
    out = m(*ins)
    w = torch.randn(out.shape, generator=gen)
    (out * w).sum().backward()
    arrays["train_b4"] = {**{f"in{i}": t.detach().numpy() for i, t in enumerate(ins)}, "out": out.detach().numpy(), "w": w.numpy(),
                          **{f"gin{i}": t.grad.numpy() for i, t in enumerate(ins)}}
    arrays["grads"] = {n: p.grad.numpy() for n, p in m.named_parameters() if p.grad is not None}
    _save(tag, **arrays)


def gen_clip(ref, tag="clip_w128_l2"):
    """CLIP VisionTransformer (forward_ori blocks), small: 4x2 patches, width 128, 2 layers."""
    torch.manual_seed(3)
    gen = torch.Generator().manual_seed(5)
    cfg = SimpleNamespace(MODEL=SimpleNamespace(PROMPT=False, ADAPTER=False))
    vt = ref.clip.VisionTransformer(4, 2, 16, 16, 128, 2, 2, 64, cfg).eval()
    with torch.no_grad():
        for n, p in vt.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
            if "ln_" in n and n.endswith("weight"):
                p.copy_(1 + torch.randn(p.shape, generator=gen) * 0.05)
    img = torch.randn(2, 3, 64, 32, generator=gen)
    cv = torch.randn(2, 1, 128, generator=gen) * 0.5
    with torch.no_grad():
        out = vt(img, cv)
        out_nocv = vt(img, None)
    _save(tag, sd=_np(vt.state_dict()), img=img, cv=cv, out=out, out_nocv=out_nocv)


def gen_vit(ref, tag="vit_w128_l2"):
    """ImageNet-style ViT `Trans` (eval): 4x2 patches, dim 128, depth 2, 2 heads, 3 cameras."""
    torch.manual_seed(9)
    gen = torch.Generator().manual_seed(13)
    from functools import partial
    m = ref.vit.Trans(img_size=(64, 32), patch_size=16, stride_size=16, embed_dim=128, depth=2, num_heads=2,
                      mlp_ratio=4, qkv_bias=True, camera=3, view=0, drop_path_rate=0.1,
                      norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), sie_xishu=1.5).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
            if "norm" in n and n.endswith("weight"):
                p.copy_(1 + torch.randn(p.shape, generator=gen) * 0.05)
    img = torch.randn(2, 3, 64, 32, generator=gen)
    cam = torch.tensor([2, 0])
    with torch.no_grad():
        out = m(img, cam_label=cam, view_label=None)
    _save(tag, sd=_np(m.state_dict()), img=img, cam=cam, out=out)


def gen_pife(ref, tag="pife_c64"):
    """rgb_reduce = LayerNorm(2C) -> Linear(2C,C) -> QuickGELU applied to cat[cls, mean(patches)]
    (make_model.py:52-58,301-303).  Built from torch modules + the reference's QuickGELU."""
    torch.manual_seed(21)
    C = 64
    red = torch.nn.Sequential(torch.nn.LayerNorm(2 * C), torch.nn.Linear(2 * C, C), ref.attnmoe.QuickGELU())
    with torch.no_grad():
        red[0].weight.add_(torch.randn(2 * C) * 0.05)
        red[0].bias.add_(torch.randn(2 * C) * 0.05)
    cls, patches = torch.randn(3, C), torch.randn(3, 8, C)
    pool = torch.nn.AdaptiveAvgPool1d(1)
    with torch.no_grad():
        out = red(torch.cat([cls, pool(patches.permute(0, 2, 1)).squeeze(-1)], -1))
    _save(tag, sd=_np(red.state_dict()), cls=cls, patches=patches, out=out)


def gen_losses(ref, tag="losses"):
    torch.manual_seed(31)
    B, K, D = 16, 11, 48
    score, feat = torch.randn(B, K), torch.randn(B, D)
    target = torch.arange(4).repeat_interleave(4)
    xent = ref.softmax.CrossEntropyLabelSmooth(num_classes=K, use_gpu=False)
    tri = ref.triplet.TripletLoss()
    _save(tag, score=score, feat=feat, target=target, xent=xent(score, target), triplet=tri(feat, target)[0])


def gen_eval(tag="evalfunc"):
    try:
        met = load_reference_metrics()
    except Exception as e:                                # plotting deps missing in a way stubs cannot cover
        print("skip eval fixture:", repr(e))
        return
    rng = np.random.RandomState(0)
    nq, ng, D = 30, 200, 32
    ids = 10
    q_p, g_p = rng.randint(0, ids, nq), rng.randint(0, ids, ng)
    q_c, g_c = rng.randint(0, 2, nq), rng.randint(0, 2, ng)
    cent = rng.randn(ids, D)
    feats = torch.tensor(np.concatenate([cent[q_p], cent[g_p]]) + rng.randn(nq + ng, D) * 1.5, dtype=torch.float32)
    ev = met.R1_mAP_eval(nq, max_rank=50, feat_norm=True)
    ev.update((feats, np.concatenate([q_p, g_p]), np.concatenate([q_c, g_c]), [""] * (nq + ng)))
    cmc, mAP = ev.compute()[:2]
    _save(tag, feats=feats, pids=np.concatenate([q_p, g_p]), camids=np.concatenate([q_c, g_c]), nq=np.asarray(nq),
          cmc=cmc, mAP=np.asarray(mAP))


def _toy_dataset(rng, n_ids, lo, hi):
    """(path, pid, camid, viewid) tuples as data/datasets/bases.py yields them; identities interleaved, some with fewer than K samples."""
    items = []
    for pid in rng.permutation(n_ids):
        for j in range(int(rng.randint(lo, hi + 1))):
            items.append(("", int(pid) * 3 + 1, int(j % 4), 0))
    order = rng.permutation(len(items))
    return [items[i] for i in order]


def gen_sampler(tag="sampler_legacy"):
    """Index sequences of the reference's OWN sampler classes (data/datasets/sampler.py:8-67, sampler_ddp.py:117-201) for seeded global
    `random` / `np.random` states (train_net.py:18-30 seeds both with SOLVER.SEED): two epochs of the single-process sampler, and the
    data-parallel sampler's global list plus every rank's part for world sizes 1, 2 and 4.  The data-parallel class asks
    torch.distributed for rank / world size in its constructor, so it is built inside a one-rank gloo group and the (rank, world,
    mini-batch) attributes are then set to each partition in turn -- its own __fetch_current_node_idxs does the cutting."""
    import random
    import tempfile
    import torch.distributed as dist
    sm = _load("ref_sampler", "data/datasets/sampler.py")
    sd = _load("ref_sampler_ddp", "data/datasets/sampler_ddp.py")
    data = _toy_dataset(np.random.RandomState(5), 23, 2, 11)
    out = {"pids": np.asarray([d[1] for d in data]), "seed": np.asarray(1234)}
    for (bs, k) in ((16, 4), (24, 8)):
        random.seed(1234)
        np.random.seed(1234)
        s = sm.RandomIdentitySampler(data, bs, k)
        out[f"single_b{bs}_k{k}_len"] = np.asarray(len(s))
        out[f"single_b{bs}_k{k}_e0"] = np.asarray([int(i) for i in s])
        out[f"single_b{bs}_k{k}_e1"] = np.asarray([int(i) for i in s])          # second epoch: the global states have moved on
    with tempfile.TemporaryDirectory() as tmp:
        dist.init_process_group("gloo", init_method="file://" + os.path.join(tmp, "pg"), rank=0, world_size=1)
        try:
            for (bs, k) in ((16, 4), (32, 4)):
                for world in (1, 2, 4):
                    s = sd.RandomIdentitySampler_DDP(data, bs, k)
                    s.world_size = world
                    s.mini_batch_size = bs // world
                    s.num_pids_per_batch = s.mini_batch_size // k
                    np.random.seed(1234)
                    seed = int(np.random.randint(2 ** 31))                     # shared_random_seed() of a one-rank group (:103-114)
                    np.random.seed(seed)
                    full = [int(i) for i in s.sample_list()]
                    out[f"ddp_b{bs}_k{k}_w{world}_seed"] = np.asarray(seed)
                    out[f"ddp_b{bs}_k{k}_w{world}_global"] = np.asarray(full)
                    length = int(np.ceil(len(full) * 1.0 / world))
                    for r in range(world):
                        s.rank = r
                        part = s._RandomIdentitySampler_DDP__fetch_current_node_idxs(full, length)
                        out[f"ddp_b{bs}_k{k}_w{world}_r{r}"] = np.asarray([int(i) for i in part])
            # the class's own __iter__ end to end (world 1), second call included
            np.random.seed(77)
            s = sd.RandomIdentitySampler_DDP(data, 16, 4)
            out["ddp_iter_e0"] = np.asarray([int(i) for i in s])
            out["ddp_iter_e1"] = np.asarray([int(i) for i in s])
        finally:
            dist.destroy_process_group()
    _save(tag, **out)


def gen_eval_msrv(tag="evalfunc_msrv"):
    """eval_func_msrv / R1_mAP (utils/metrics.py:12-107,172-218): the MSVR310 protocol (same identity AND same scene removed).  The
    reference writes a rank list to ./re.txt while it evaluates, so this runs inside a temporary directory."""
    import tempfile
    try:
        met = load_reference_metrics()
    except Exception as e:
        print("skip msrv fixture:", repr(e))
        return
    rng = np.random.RandomState(3)
    nq, ng, D, ids = 40, 260, 32, 12
    q_p, g_p = rng.randint(0, ids, nq), rng.randint(0, ids, ng)
    q_c, g_c = rng.randint(0, 8, nq), rng.randint(0, 8, ng)
    q_s, g_s = rng.randint(0, 3, nq), rng.randint(0, 3, ng)
    cent = rng.randn(ids, D)
    feats = torch.tensor(np.concatenate([cent[q_p], cent[g_p]]) + rng.randn(nq + ng, D) * 1.6, dtype=torch.float32)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            ev = met.R1_mAP(nq, max_rank=50, feat_norm='yes')
            ev.reset()
            ev.update((feats, np.concatenate([q_p, g_p]), np.concatenate([q_c, g_c]), np.concatenate([q_s, g_s]), [""] * (nq + ng)))
            cmc, mAP = ev.compute()[:2]
        finally:
            os.chdir(cwd)
    _save(tag, feats=feats, pids=np.concatenate([q_p, g_p]), camids=np.concatenate([q_c, g_c]), sceneids=np.concatenate([q_s, g_s]),
          nq=np.asarray(nq), cmc=cmc, mAP=np.asarray(mAP))


def main():
    if len(sys.argv) > 1:                                  # e.g. `python oracle/gen_golden.py gen_sampler gen_eval_msrv`
        for name in sys.argv[1:]:
            globals()[name]()
        return
    ref = load_reference()
    gen_fusion(ref)
    gen_fusion(ref, C=64, H=2, n_tok=5, tag="fusion_c64_h2", with_grads=True)
    gen_clip(ref)
    gen_vit(ref)
    gen_pife(ref)
    gen_losses(ref)
    gen_eval()
    gen_dgaf()
    gen_sampler()
    gen_eval_msrv()


if __name__ == "__main__":
    main()
