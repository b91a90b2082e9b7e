"""Where one cfg5 trial spends its time: a 3-epoch DirectPred trial at cfg2 size (2 x 20000 features,
N = 2048, B = 128), phase by phase; every phase boundary synchronises the device.  Writes a markdown table to stdout."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = {"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "batch_size": B, "epochs": 3}
tr, va = split_indices(2048, 0.2, 0)
rows = []
for trial in range(4):
    ph = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    torch.manual_seed(trial)
    m = DirectPred(cfg, ds, ["y"], device_type="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m._bind("cuda")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = fit(m, ds, tr, va, batch_size=B, epochs=3, lr=1e-3, seed=trial, prof=ph)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.cuda.synchronize(); t4 = time.perf_counter()
    del sd
    torch.cuda.synchronize(); t4a = time.perf_counter()
    del m
    torch.cuda.synchronize(); t4b = time.perf_counter()
    import gc; gc.collect()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    ph = {"model constructor": t1 - t0, "bind (arenas)": t2 - t1, **ph, "fit() outside the phases": (t3 - t2) - sum(ph.values()),
          "state_dict clone": t4 - t3, "del clone": t4a - t4, "del model (plans, graphs, arenas)": t4b - t4a, "gc.collect()": t5 - t4b}
    rows.append((ph, res.steps, t5 - t0))
keys = list(rows[-1][0])
print(f"| phase (B = {B}, {rows[-1][1]} steps) | " + " | ".join(f"trial {i} ms" for i in range(len(rows))) + " |")
print("|---|" + "---|" * len(rows))
for k in keys:
    print(f"| {k} | " + " | ".join(f"{r[0].get(k, 0.0) * 1e3:.1f}" for r in rows) + " |")
print("| **total** | " + " | ".join(f"{r[2] * 1e3:.1f}" for r in rows) + " |")
