"""Differential probe (two library builds, same inputs): one eval-free TRAIN forward of the fine-tune golden's DirectPred through
model._plan, then checksums of the chain's buffers."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from golden_io import FinetuneLoopGolden
from flexynesis_amd import models as M
from flexynesis_amd.data import MultiOmicDataset
G = FinetuneLoopGolden()
spec = G.spec
dat, ann = G.sub("dat"), G.sub("ann")
vt = {v: ("categorical" if kind == "categorical" else "numerical") for (v, kind, _) in spec.variables}
feats = {k: [f"{k}_{j}" for j in range(v.shape[1])] for k, v in dat.items()}
ds = MultiOmicDataset(dict(dat), dict(ann), vt, feats, [f"s{i}" for i in range(G.n)], {})
cfg = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": G.lrs[0],
       "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": G.max_epoch, "batch_size": G.B}
m = M.DirectPred(cfg, ds, [v[0] for v in spec.variables], device_type="cuda")
m.load_state_dict(G.sub("state0"))
B = 10
from flexynesis_amd.engine import StepPlan
for train in (True, False):
    if train:       # supplied draws: every dropout mask all ones -> the same inputs under every build
        plan = StepPlan(m._bind(), B, train=True, fused=False, supplied_draws=True, seed=0, forward_alone=True)
        plan.set_draws({k: torch.ones_like(v) for k, v in plan.draws.items()})
    else:
        plan = m._plan(B, train=train)
    x_list = [torch.as_tensor(dat[k][:B]).float() for k in dat]
    plan.set_batch(x_list=[x.to(plan.dev) for x in x_list], y=None)
    for t in plan.y.values():
        t.fill_(float("nan"))
    plan.forward()
    torch.cuda.synchronize()
    print("train" if train else "eval")
    for k in sorted(plan.buf):
        v = plan.buf[k]
        if torch.is_tensor(v) and v.is_floating_point() and any(s in k for s in ("parts", "emb", "ecat", "/a1", "/y1", "out")):
            vv = v.float()
            print(f"   {k:40s} {tuple(v.shape)}  sum {float(vv.double().sum()):+.7f}  abs {float(vv.double().abs().sum()):.7f}  colsum0 {[round(float(c), 5) for c in vv.reshape(-1, vv.shape[-1]).double().sum(0)[:8].tolist()]}")
    print("   emb", [round(float(c), 5) for c in plan.embeddings.double().sum(0).tolist()])
    if train:
        for k in ("encoders.0/layer_out_parts", "ecat", "emb"):
            print("   FULL", k, [[round(float(c), 4) for c in row] for row in plan.buf[k].reshape(-1, plan.buf[k].shape[-1])[:3].tolist()])
