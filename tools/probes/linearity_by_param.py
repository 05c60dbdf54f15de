"""Which parameters break `two backward passes = twice the gradient` (tests/test_gpu_train.py, bench-size config A, bf16)?
Prints the per-parameter relative deviation of g2 from 2 g1, largest first.  python tools/probes/linearity_by_param.py [passes]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from demo2_amd.modeling import make_model
from demo2_amd.modeling.hip_backward import TrainStep
from oracle import demo_oracle as O
from util import make_cfg, spec_of, synth_batch

DEV = "cuda"
cfg = make_cfg("RGBNT201")
if os.environ.get("LIN_TRIPLET") is not None:            # the batch-hard triplet loss picks arg max / arg min pairs: near-ties can flip between passes
    cfg.defrost() if hasattr(cfg, "defrost") else None
    cfg.MODEL.TRIPLET_LOSS_WEIGHT = float(os.environ["LIN_TRIPLET"])
torch.manual_seed(0)
m = make_model(cfg, 171, 4, 0)
m.load_state_dict(O.init_state_dict(spec_of(m), seed=1))
m = m.to(DEV).train().set_compute_dtype(torch.bfloat16)
m.geom.hdm_dropout = 0.0
m.geom.drop_path = None
x, cam = synth_batch(128, 256, 128, 4, seed=21)
xd, camd = {k: v.to(DEV) for k, v in x.items()}, cam.to(DEV)
target = (torch.arange(128, device=DEV) // 4) % 171
ts = TrainStep(m, cfg, 1, optimizer=False)
gs = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    l = ts.step(xd, target, camd).item()
    gs.append(m.grad_buffer().clone())
    print("pass", it, "loss", l)
torch.cuda.synchronize()
inc = [gs[0]] + [gs[i] - gs[i - 1] for i in range(1, len(gs))]
for i in range(1, len(inc)):
    print(f"pass {i} increment vs pass 0: rel {((inc[i] - inc[0]).norm() / inc[0].norm()).item():.3e}")
# pairwise: which passes agree (two levels show up: a ~2e-3 floor and a ~9e-3 "other mode")
for i in range(len(inc)):
    print("pairwise", i, " ".join(f"{((inc[i] - inc[j]).norm() / inc[0].norm()).item():.1e}" for j in range(len(inc))))
far = max(range(1, len(inc)), key=lambda i: (inc[i] - inc[0]).norm().item())
print("per-parameter: pass", far, "against pass 0")
rows = []
for name, (o, shape, n) in m.ptab.index.items():
    a, b = inc[0][o:o + n], inc[far][o:o + n]
    if a.norm() > 0:
        rows.append((((b - a).norm() / inc[0].norm()).item(), ((b - a).norm() / a.norm()).item(), name, n))
rows.sort(reverse=True)
sel = os.environ.get("LIN_FILTER")
for r in (rows if sel else rows[:25]):
    if not sel or sel in r[2]:
        print(f"{r[0]:.3e} of the whole  {r[1]:.3e} of its own  {r[2]}  ({r[3]})")
