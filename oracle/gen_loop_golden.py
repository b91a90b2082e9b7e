"""Generate tests/golden/trial_loop_directpred.npz FROM THE REFERENCE's model class (build container only):

    python -m oracle.gen_loop_golden

One small DirectPred trial (2 omics, regression + classification + survival heads, uncertainty weighting, NaN / -1
labels) driven through the schedule of oracle/loop.py -- per epoch: the reference's own ``training_step`` (log=True, the
logged dict captured) -> backward -> clip -> Adam for every full batch of a recorded shuffle, then the reference's
``validation_step`` over the validation split in order.  Stored: inputs (spec, initial state, cohort, split, shuffles,
every dropout mask) and what the reference produced (per-epoch means of every logged value, per-epoch validation loss,
final state).  Early stopping / epoch reduction are Lightning's (absent): the golden records the per-batch values so the
restated reduction can be checked against them, not Lightning itself."""
from __future__ import annotations

import dataclasses
import json
import os

import numpy as np
import torch

from . import ref_capture, ref_shim
from .gen_goldens import make_cohort, perturbed_state
from .restate import Spec

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "trial_loop_directpred.npz")


def main():
    R = ref_shim.load()
    spec = Spec("DirectPred", [("gex", 40), ("cnv", 28)], 6, 0.3, 4,
                [("y", "numerical", 1), ("c", "categorical", 3), ("event", "numerical", 1)],
                surv_event_var="event", surv_time_var="time")
    n, B, epochs, lr = 58, 8, 7, 1e-3
    dat, ann, vt = make_cohort(spec, n, seed=21, missing=True)
    ds = ref_capture.make_dataset(R, dat, ann, vt)
    cfg = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": lr,
           "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": epochs, "batch_size": B}
    model = ref_capture.build_reference_model(R, spec, ds, cfg)
    st0 = perturbed_state(spec, seed=8)
    model.load_state_dict(st0)
    g = torch.Generator().manual_seed(5)
    split = torch.randperm(n, generator=g)
    n_val = int(n * 0.2)                                   # main.py:272-276
    train_idx, val_idx = split[: n - n_val], split[n - n_val:]
    nb = train_idx.numel() // B
    perms = [torch.randperm(train_idx.numel(), generator=g) for _ in range(epochs)]
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    logged = {}
    model.log_dict = lambda d, *a, **k: logged.update({kk: float(torch.as_tensor(v).detach().reshape(-1)[0]) for kk, v in d.items()})
    keys = [v[0] for v in spec.variables] + [spec.surv_time_var]
    out = {"spec_json": json.dumps(dataclasses.asdict(spec)), "lr": lr, "epochs": epochs, "batch_size": B,
           "train_idx": train_idx.numpy(), "val_idx": val_idx.numpy()}
    for k, v in st0.items():
        out[f"state0/{k}"] = v.numpy()
    for k, v in dat.items():
        out[f"dat/{k}"] = v.numpy()
    for k, v in ann.items():
        out[f"ann/{k}"] = v.numpy()

    def batch(rows):
        return {"x": [dat[name][rows] for name, _ in spec.layers], "y": {k: ann[k][rows] for k in keys}}

    for e in range(epochs):
        out[f"perm/{e}"] = perms[e].numpy()
        model.train()
        perm = train_idx[perms[e]]
        for b in range(nb):
            rows = perm[b * B:(b + 1) * B]
            opt.zero_grad()
            logged.clear()
            with ref_capture.capture_rng() as cap:
                loss = model.training_step(ref_capture.reference_batch(spec, batch(rows)), b, log=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            for k, v in ref_capture.name_draws(spec, cap).items():
                out[f"draws/{e}/{b}/{k}"] = v.numpy()
            for k, v in logged.items():
                out[f"step/{e}/{b}/{k}"] = np.float64(v)
        for k, v in model.state_dict().items():                  # the weights this epoch's validation sees
            out[f"state_epoch/{e}/{k}"] = v.detach().numpy().copy()
        model.eval()
        with torch.no_grad():
            for bi, s in enumerate(range(0, val_idx.numel(), B)):
                rows = val_idx[s:s + B]
                logged.clear()
                vl = model.validation_step(ref_capture.reference_batch(spec, batch(rows)), bi, log=True)
                out[f"val/{e}/{bi}/val_loss"] = np.float64(float(torch.as_tensor(vl).reshape(-1)[0]))
                out[f"val/{e}/{bi}/n"] = np.int64(rows.numel())
                assert abs(logged["val_loss"] - float(torch.as_tensor(vl).reshape(-1)[0])) < 1e-12
    for k, v in model.state_dict().items():
        out[f"state_final/{k}"] = v.detach().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
