import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from golden_io import FinetuneLoopGolden
from flexynesis_amd import models as M
from flexynesis_amd.data import MultiOmicDataset
from flexynesis_amd.fit import fine_tune
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
def supplied_for(unit):
    pf = G.perms_fn(unit)
    return {"perms": [pf(e) for e in range(G.max_epoch)], "draws": G.draws_fn(unit)}
details = {}
import flexynesis_amd.engine as E
_adv = E.PipelinedStep._advance
_cnt = [0]
def adv_dbg(self):
    plan = self.plans[self.k]
    if _cnt[0] < 3:
        torch.cuda.synchronize()
        for k in ("encoders.0/a1", "encoders.0/layer_out_parts", "encoders.1/layer_out_parts", "ecat", "emb", "MLPs.y/y1", "MLPs.y/save_mean", "MLPs.c/y1", "MLPs.c/save_mean"):
            if k in plan.buf:
                v = plan.buf[k].double()
                print("STEP", _cnt[0], k, tuple(v.shape), round(float(v.sum()), 6), round(float(v.abs().sum()), 6), [round(float(c), 5) for c in v.reshape(-1, v.shape[-1]).sum(0)[:6].tolist()], flush=True)
        print("STEP", _cnt[0], "bufkeys", [k for k in plan.buf if "save" in k or "MLPs" in k][:20], flush=True)
    _cnt[0] += 1
    return _adv(self)
E.PipelinedStep._advance = adv_dbg
import flexynesis_amd.fit as F
_fit = F.fit
def fit_dbg(*a, **k):
    r = _fit(*a, **k)
    sd = a[0].state_dict()
    print("RM", {kk: [round(float(x), 5) for x in vv.flatten().tolist()] for kk, vv in sd.items() if "MLPs" in kk and "running" in kk}, flush=True)
    print("STATE", {kk: round(float(vv.double().abs().sum()), 7) for kk, vv in sd.items() if "running" in kk or "batchnorm" in kk}, flush=True)
    print("FIT", k.get("frozen"), [{kk: round(float(vv), 6) for kk, vv in h.items()} for h in r.history][:5], flush=True)
    return r
F.fit = fit_dbg
final, best, results = fine_tune(m, ds, n_splits=G.n_splits, batch_size=G.B, learning_rates=G.lrs[:1], max_epoch=G.max_epoch,
                                 freeze_configs=G.cfgs[:1], seed=G.kfold_seed, device="cuda", use_graph=False,
                                 supplied_for=supplied_for, details=details)
for k, v in details.items():
    print(k, {kk: (vv if not torch.is_tensor(vv) else vv.tolist()) for kk, vv in v.items() if kk not in ("state",)} if isinstance(v, dict) else v)
