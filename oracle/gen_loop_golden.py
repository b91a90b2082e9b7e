"""Generate tests/golden/trial_loop_{directpred,supervised_vae,triplet}.npz FROM THE REFERENCE's model classes (build
container only):

    python -m oracle.gen_loop_golden [directpred] [supervised_vae] [triplet]

One small trial per model class (DirectPred: 2 omics, regression + classification + survival heads, uncertainty weighting,
NaN / -1 labels; supervised_vae: eps / MMD-prior draws recorded in training AND validation; MultiTripletNetwork: anchors =
samples with a main label, the positives / negatives the reference's TripletMultiOmicDataset drew) driven through the schedule of oracle/loop.py -- per epoch: the reference's own ``training_step`` (log=True, the
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

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
OUT = os.path.join(GOLDEN_DIR, "trial_loop_directpred.npz")

CASES = {
    # name -> (file, spec, n, B, epochs, cohort seed, state seed)
    "directpred": ("trial_loop_directpred.npz",
                   Spec("DirectPred", [("gex", 40), ("cnv", 28)], 6, 0.3, 4,
                        [("y", "numerical", 1), ("c", "categorical", 3), ("event", "numerical", 1)],
                        surv_event_var="event", surv_time_var="time"), 58, 8, 7, 21, 8),
    # the VAE's epoch means of mmd_loss and its validation sum (supervised_vae.py:338-381): eps / prior draws recorded for
    # validation batches too (the reference samples z in eval mode as well, supervised_vae.py:417-419)
    "supervised_vae": ("trial_loop_supervised_vae.npz",
                       Spec("supervised_vae", [("gex", 36), ("cnv", 24)], 6, 0.3, 4,
                            [("y", "numerical", 1), ("c", "categorical", 3)]), 50, 8, 5, 31, 12),
    # the triplet network: anchors are the samples with a non-NaN main label ("n = valid anchors", data.py:1102-1104,
    # main.py:176-181), positives / negatives as drawn by the reference's TripletMultiOmicDataset.__getitem__ (recorded)
    "triplet": ("trial_loop_triplet.npz",
                Spec("MultiTripletNetwork", [("gex", 40), ("cnv", 32)], 6, 0.25, 4,
                     [("c", "categorical", 3), ("y", "numerical", 1)]), 64, 8, 5, 41, 16),
}


class _RecordChoice:
    """Records what numpy.random.choice returns while the reference's TripletMultiOmicDataset.__getitem__ draws the positive
    (re-drawn until it differs from the anchor) and the negative sample: the last two values of a call are (positive, negative)."""

    def __enter__(self):
        self.orig, self.values = np.random.choice, []

        def rec(*a, **k):
            v = self.orig(*a, **k)
            self.values.append(int(v))
            return v
        np.random.choice = rec
        return self

    def __exit__(self, *exc):
        np.random.choice = self.orig
        return False


def _collate(items):
    """default_collate of MultiOmicDataset items' (dat dict, ann dict) parts (reference data.py:980-995)."""
    dat = {k: torch.stack([it[0][k] for it in items]) for k in items[0][0]}
    ann = {k: torch.stack([torch.as_tensor(it[1][k]) for it in items]) for k in items[0][1]}
    return dat, ann


def _val_draws(spec, cap):
    """eval-mode draws of the VAE family: eps and the MMD priors (no dropout in eval mode)."""
    d = {"eps": cap.randn_like[0]}
    for j in range(len(spec.dec_idx)):
        d[f"prior.{j}"] = cap.randn[j]
    return d


def generate(name: str):
    import random
    R = ref_shim.load()
    fname, spec, n, B, epochs, cseed, sseed = CASES[name]
    lr = 1e-3
    trip, vae = spec.model == "MultiTripletNetwork", spec.is_vae
    dat, ann, vt = make_cohort(spec, n, seed=cseed, missing=True)
    if name != "directpred":          # missing labels inside the cohort (the DirectPred cohort is kept as generated in round 2)
        g0 = torch.Generator().manual_seed(cseed + 1)
        for (v, kind, C) in spec.variables:
            hole = torch.randperm(n, generator=g0)[: max(n // 9, 2)]
            ann[v][hole] = float("nan")
        # the reference sizes a categorical head by len(np.unique(labels)), NaN included (direct_pred.py:97-100)
        spec = dataclasses.replace(spec, variables=[(v, kind, int(len(np.unique(ann[v].numpy()))) if kind == "categorical" else C)
                                                     for (v, kind, C) in spec.variables])
    ds = ref_capture.make_dataset(R, dat, ann, vt)
    cfg = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": lr,
           "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": epochs, "batch_size": B}
    model = ref_capture.build_reference_model(R, spec, ds, cfg)
    st0 = perturbed_state(spec, seed=sseed)
    model.load_state_dict(st0)
    g = torch.Generator().manual_seed(5)
    tds = R.TripletMultiOmicDataset(ds, spec.variables[0][0]) if trip else None
    n_items = len(tds) if trip else n                     # len(loader_dataset): the valid anchors for the triplet network
    split = torch.randperm(n_items, generator=g)
    n_val = int(n_items * 0.2)                             # main.py:272-276
    train_idx, val_idx = split[: n_items - n_val], split[n_items - n_val:]
    nb = train_idx.numel() // B
    perms = [torch.randperm(train_idx.numel(), generator=g) for _ in range(epochs)]
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    logged = {}
    model.log_dict = lambda d, *a, **k: logged.update({kk: float(torch.as_tensor(v).detach().reshape(-1)[0]) for kk, v in d.items()})
    keys = [v[0] for v in spec.variables] + ([spec.surv_time_var] if spec.surv_time_var else [])
    out = {"spec_json": json.dumps(dataclasses.asdict(spec)), "lr": lr, "epochs": epochs, "batch_size": B,
           "train_idx": train_idx.numpy(), "val_idx": val_idx.numpy()}
    if trip:
        out["valid_indices"] = np.asarray(tds.valid_indices, np.int64)
        np.random.seed(1234)
        random.seed(1234)
    for k, v in st0.items():
        out[f"state0/{k}"] = v.numpy()
    for k, v in dat.items():
        out[f"dat/{k}"] = v.numpy()
    for k, v in ann.items():
        out[f"ann/{k}"] = v.numpy()

    def batch(rows, tag):
        """The collated batch of loader items ``rows`` (and, for the triplet network, the recorded positive / negative rows)."""
        if not trip:
            return ref_capture.reference_batch(spec, {"x": [dat[name_][rows] for name_, _ in spec.layers],
                                                      "y": {k: ann[k][rows] for k in keys}})
        items, pos, neg = [], [], []
        for i in rows.tolist():
            with _RecordChoice() as rc:
                items.append(tds[i])
            pos.append(rc.values[-2])
            neg.append(rc.values[-1])
        out[f"{tag}/pos"], out[f"{tag}/neg"] = np.asarray(pos, np.int64), np.asarray(neg, np.int64)
        a = _collate([(it[0], it[3]) for it in items])
        p_ = _collate([(it[1], it[3]) for it in items])[0]
        n_ = _collate([(it[2], it[3]) for it in items])[0]
        return (a[0], p_, n_, a[1])

    def validate(e):
        model.eval()
        with torch.no_grad():
            for bi, s in enumerate(range(0, val_idx.numel(), B)):
                rows = val_idx[s:s + B]
                logged.clear()
                with ref_capture.capture_rng() as cap:
                    vl = model.validation_step(batch(rows, f"vtrip/{e}/{bi}"), bi, log=True)
                if vae:
                    for k, v in _val_draws(spec, cap).items():
                        out[f"vdraws/{e}/{bi}/{k}"] = v.numpy()
                out[f"val/{e}/{bi}/val_loss"] = np.float64(float(torch.as_tensor(vl).reshape(-1)[0]))
                out[f"val/{e}/{bi}/n"] = np.int64(rows.numel())
                assert abs(logged["val_loss"] - float(torch.as_tensor(vl).reshape(-1)[0])) < 1e-12

    for e in range(epochs):
        out[f"perm/{e}"] = perms[e].numpy()
        model.train()
        perm = train_idx[perms[e]]
        for b in range(nb):
            rows = perm[b * B:(b + 1) * B]
            opt.zero_grad()
            logged.clear()
            bt = batch(rows, f"trip/{e}/{b}")
            with ref_capture.capture_rng() as cap:
                loss = model.training_step(bt, b, log=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            for k, v in ref_capture.name_draws(spec, cap).items():
                out[f"draws/{e}/{b}/{k}"] = v.numpy()
            for k, v in logged.items():
                out[f"step/{e}/{b}/{k}"] = np.float64(v)
        for k, v in model.state_dict().items():                  # the weights this epoch's validation sees
            out[f"state_epoch/{e}/{k}"] = v.detach().numpy().copy()
        validate(e)
    if name != "directpred":
        validate(epochs)                                         # trainer.validate after fit: fresh draws / triplets (main.py:323-326)
    for k, v in model.state_dict().items():
        out[f"state_final/{k}"] = v.detach().numpy()
    path = os.path.join(GOLDEN_DIR, fname)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main(argv=None):
    import sys
    names = (argv if argv is not None else sys.argv[1:]) or ["supervised_vae", "triplet"]   # (directpred: round 2's file, kept)
    for nm in names:
        generate(nm)


if __name__ == "__main__":
    main()
