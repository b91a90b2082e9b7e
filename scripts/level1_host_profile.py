"""Where the HOST spends a level-1 (Lightning protocol) training step: enqueue time per step against drained time per step, then cProfile of 200 steps.
python scripts/level1_host_profile.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
from flexynesis_amd import models as M
from flexynesis_amd.data import MultiOmicDataset
dev = torch.device("cuda:0")
n, F, B = 2048, 20000, 128
g = torch.Generator(device=dev); g.manual_seed(1234)
dat = {k: torch.randn(n, F, generator=g, device=dev) for k in ("gex", "cnv")}
ann = {"y": dat["gex"][:, :16].sum(1) / 4 + 0.1 * torch.randn(n, generator=g, device=dev)}
feats = {k: [f"{k}_{j}" for j in range(F)] for k in dat}
ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(n)], {})
cfg = {"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "epochs": 1, "batch_size": B}
m = M.DirectPred(cfg, ds, ["y"], device_type="cuda"); m.to(dev); m.train(); m.fused_optimizer = True
opt = m.configure_optimizers()
perm = torch.randperm(n, generator=g, device=dev)
def step(i):
    o = (i * B) % (n - B); idx = perm[o:o + B]
    batch = ({k: v[idx] for k, v in dat.items()}, {"y": ann["y"][idx]}, None)
    opt.zero_grad()
    loss = m.training_step(batch, i, log=False)
    loss.backward()
    m.configure_gradient_clipping(opt, 1.0, "norm")
    opt.step()
    return loss
for i in range(10): step(i)
torch.cuda.synchronize()
# host-only cost: how long does enqueueing one step take when the GPU is not waited for?
t0 = time.perf_counter()
for i in range(200): step(10 + i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 200:.3f} ms/step, with drain {1e3 * (t2 - t0) / 200:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for i in range(200): step(300 + i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
