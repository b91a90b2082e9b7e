"""Level-1 drop-in throughput: the Lightning protocol (zero_grad -> training_step -> loss.backward() -> clip_grad_norm_ ->
optimizer.step()) on the model class with torch-side batches, at cfg2 (2 x 20000 features, B = 128), against the engine
loop (fit / bench.py).  Gradients are materialised (36 B/param/step instead of 24) and torch's clip_grad_norm_ makes two
more passes over them; this is the price of running under an unmodified external loop.

    python scripts/bench_dropin.py [fx|torch|fused]

"fused": model.fused_optimizer = True -- FxAdam runs the engine's clip + dW+Adam launches (24 B/param, no wide gradients); the
clip value reaches it through the configure_gradient_clipping hook, as under Lightning's Trainer."""
import json
import sys
import time

sys.path.insert(0, ".")
import torch

from flexynesis_amd import models as M
from flexynesis_amd.data import MultiOmicDataset


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fx"
    dev = torch.device("cuda:0")
    n, F, B = 2048, 20000, 128
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    dat = {k: torch.randn(n, F, generator=g, device=dev) for k in ("gex", "cnv")}
    ann = {"y": dat["gex"][:, :16].sum(1) / 4 + 0.1 * torch.randn(n, generator=g, device=dev)}
    feats = {k: [f"{k}_{j}" for j in range(F)] for k in dat}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(n)], {})
    cfg = {"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "epochs": 1, "batch_size": B}
    m = M.DirectPred(cfg, ds, ["y"], device_type="cuda")
    m.to(dev)
    m.train()
    m.fused_optimizer = kind == "fused"
    opt = m.configure_optimizers() if kind in ("fx", "fused") else torch.optim.Adam(m.parameters(), lr=1e-3)
    perm = torch.randperm(n, generator=g, device=dev)

    def step(i):
        idx = perm[(i * B) % (n - B):(i * B) % (n - B) + B]
        batch = ({k: v[idx] for k, v in dat.items()}, {"y": ann["y"][idx]}, None)
        opt.zero_grad()
        loss = m.training_step(batch, i, log=False)
        loss.backward()
        m.configure_gradient_clipping(opt, 1.0, "norm")      # Lightning's hook: torch clip_grad_norm_ unless the optimiser clips itself
        opt.step()
        return loss

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 40
    for i in range(K):
        loss = step(5 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"path": "Lightning protocol (level-1 drop-in)", "optimizer": type(opt).__name__ + (" (fused)" if getattr(opt, "fused", False) else ""), "ms_per_step": round(1e3 * dt / K, 3),
                      "samples_per_s": round(K * B / dt, 1), "loss": float(loss)}))


if __name__ == "__main__":
    main()
