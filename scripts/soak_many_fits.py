"""Stress: many fits in one process (HPO sweeps and the FineTuner run dozens): 45-fit fine_tune + a 24-trial sweep."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd.data import MultiOmicDataset
from flexynesis_amd.models import DirectPred, supervised_vae
from flexynesis_amd.fit import fine_tune
from flexynesis_amd import sweep
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
dat = {"gex": torch.randn(300, 1500, generator=g), "cnv": torch.randn(300, 1100, generator=g)}
ann = {"y": dat["gex"][:, :8].sum(1) + 0.1 * torch.randn(300, generator=g), "c": (dat["cnv"][:, 0] > 0).float()}
ds = MultiOmicDataset(dat, ann, {"y": "numerical", "c": "categorical"}, {k: [f"{k}{i}" for i in range(v.shape[1])] for k, v in dat.items()},
                      [f"s{i}" for i in range(300)], {})
cfg = {"latent_dim": 32, "hidden_dim_factor": 0.8, "lr": 1e-3, "supervisor_hidden_dim": 16, "epochs": 3, "batch_size": 32}
for cls in (DirectPred, supervised_vae):
    t0 = time.perf_counter()
    m = cls(cfg, ds, ["y", "c"], device_type="cuda")
    final, best, results = fine_tune(m, ds, n_splits=5, batch_size=32, max_epoch=6, seed=1, device="cuda")
    print(cls.__name__, "fine_tune:", len(results), "configs x 5 folds in", round(time.perf_counter() - t0, 1), "s; best", best["freeze"],
          best["learning_rate"], round(best["average_val_loss"], 4), flush=True)
    assert all(np.isfinite(r["average_val_loss"]) for r in results)
sweep.main(["--trials", "24", "--epochs", "2", "--features", "6000", "--samples", "1024"])
print("soak ok")
