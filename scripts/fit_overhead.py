"""Fixed vs per-step cost of fit() (plan construction, graph capture, validation) at cfg2 shapes."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd.data import MultiOmicDataset
from flexynesis_amd.models import DirectPred
from flexynesis_amd.fit import fit, split_indices
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
dat = {k: torch.randn(2048, 20000, generator=g, device=dev) for k in ("gex", "cnv")}
ann = {"y": dat["gex"][:, :16].sum(1) / 4}
feats = {k: [f"{k}_{i}" for i in range(20000)] for k in dat}
ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(2048)], {})
cfg = {"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "batch_size": 128, "epochs": 3}
tr, va = split_indices(2048, 0.2, 0)
for ep in (1, 3, 3, 30):
    m = DirectPred(cfg, ds, ["y"], device_type="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = fit(m, ds, tr, va, batch_size=128, epochs=ep, lr=1e-3, seed=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epochs {ep:3d}: {dt*1e3:8.1f} ms total, {res.steps} steps -> {dt*1e3/res.steps:6.2f} ms/step all-in", flush=True)
