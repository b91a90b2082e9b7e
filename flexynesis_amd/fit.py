"""Epoch loop over a device-resident cohort: the engine-side counterpart of
``trainer.fit`` + ``trainer.validate`` inside ``HyperparameterTuning.objective`` (reference
main.py:228-333) with the Trainer configuration of main.py:212-225 (max_epochs, clip 1.0 "norm",
EarlyStopping(val_loss, patience, mode=min), train loader shuffle=True drop_last=True, validation loader
in order).  Every optimisation step is one hipGraph replay of the recorded HIP tapes; the only host
work per epoch is drawing a permutation on the device and reading back a handful of scalars.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .data import DeviceCohort
from .engine import PipelinedStep, StepPlan


@dataclass
class FitResult:
    val_loss: float
    epochs_run: int
    stopped_epoch: int                      # 0 if early stopping never fired (Lightning's convention)
    history: List[Dict[str, float]] = field(default_factory=list)
    steps: int = 0


def _cohort_of(dataset, device) -> DeviceCohort:
    c = getattr(dataset, "_fx_cohort", None)
    if c is None or c.device != torch.device(device):
        c = DeviceCohort.from_dataset(dataset, device)
        try:
            dataset._fx_cohort = c
        except Exception:
            pass
    return c


class TripletSampler:
    """On-device version of TripletMultiOmicDataset.__getitem__'s sampling (reference data.py:1106-1131):
    positive = uniformly random OTHER sample with the anchor's label; negative = uniformly random member of a
    uniformly chosen other label group, NaN labels forming one extra group."""

    def __init__(self, labels: torch.Tensor):
        dev = labels.device
        lab = labels.float()
        nan = torch.isnan(lab)
        uniq = torch.unique(lab[~nan])
        gid = torch.bucketize(torch.where(nan, uniq[0] if uniq.numel() else lab.new_zeros(()), lab), uniq)
        self.n_groups = int(uniq.numel())
        if bool(nan.any()):
            gid = torch.where(nan, torch.full_like(gid, self.n_groups), gid)
            self.n_groups += 1
        self.gid = gid
        self.order = torch.argsort(gid, stable=True)
        counts = torch.bincount(gid, minlength=self.n_groups)
        self.counts = counts
        self.starts = torch.cumsum(counts, 0) - counts
        rank = torch.empty_like(self.order)
        rank[self.order] = torch.arange(gid.numel(), device=dev)
        self.rank_in_group = rank - self.starts[gid]
        self.valid = torch.nonzero(~nan).reshape(-1)
        if self.n_groups < 2:
            raise ValueError("triplet sampling needs at least two label groups")

    def sample(self, anchors: torch.Tensor, gen: torch.Generator):
        g = self.gid[anchors]
        cnt = self.counts[g]
        if bool((cnt < 2).any()):
            raise ValueError("a label group has a single member: no positive sample exists for its anchor")
        u = torch.rand(anchors.shape, generator=gen, device=anchors.device)
        r = torch.minimum((u * (cnt - 1).float()).long(), cnt - 2)
        r = torch.where(r >= self.rank_in_group[anchors], r + 1, r)
        pos = self.order[self.starts[g] + r]
        u2 = torch.rand(anchors.shape, generator=gen, device=anchors.device)
        og = torch.minimum((u2 * (self.n_groups - 1)).long(), torch.full_like(g, self.n_groups - 2))
        og = torch.where(og >= g, og + 1, og)
        u3 = torch.rand(anchors.shape, generator=gen, device=anchors.device)
        ocnt = self.counts[og]
        rn = torch.minimum((u3 * ocnt.float()).long(), ocnt - 1)
        neg = self.order[self.starts[og] + rn]
        return pos, neg


def _eval_loss(model, store, cohort, idx_rows: torch.Tensor, batch_size: int, passes: int, sampler, gen, cache) -> float:
    """Mean validation total (batch-size weighted, like Lightning's epoch reduction of validation_step)."""
    n = idx_rows.numel()
    tot, cnt = 0.0, 0
    acc = []
    for s in range(0, n, batch_size):
        rows = idx_rows[s:s + batch_size]
        B = rows.numel()
        if B not in cache:
            cache[B] = StepPlan(store, B, train=False, cohort=cohort, n_batches=0, seed=model._seed + 7919 + B)
        plan = cache[B]
        if passes == 3:
            pos, neg = sampler.sample(rows, gen)
            plan.idx.copy_(torch.cat([rows, pos, neg]))
        else:
            plan.idx.copy_(rows)
        plan.t_gather.run()
        plan.forward()
        k = len(plan.spec.loss_names())
        acc.append((plan.loss_vec[k].clone(), B))
    for v, B in acc:
        tot += float(v) * B
        cnt += B
    return tot / max(cnt, 1)


def fit(model, dataset, train_idx: Sequence[int], val_idx: Optional[Sequence[int]] = None, *, batch_size: int,
        epochs: int, lr: float, patience: int = 0, seed: int = 0, use_graph: bool = True, device=None,
        verbose: bool = False) -> FitResult:
    """Train ``model`` on ``dataset[train_idx]`` and validate on ``dataset[val_idx]`` once per epoch.
    For MultiTripletNetwork the indices address the valid (non-NaN main label) anchors, like the reference's
    ``TripletMultiOmicDataset`` (data.py:1102-1104)."""
    store = model._bind(device)
    dev = store.device
    cohort = _cohort_of(dataset, dev)
    spec = model.spec
    trip = spec.model == "MultiTripletNetwork"
    passes = 3 if trip else 1
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    sampler = None
    tr = torch.as_tensor(np.asarray(train_idx), dtype=torch.int64, device=dev)
    va = torch.as_tensor(np.asarray(val_idx), dtype=torch.int64, device=dev) if val_idx is not None and len(val_idx) else None
    if trip:
        sampler = TripletSampler(cohort.ann[model.main_var])
        tr = sampler.valid[tr]
        va = sampler.valid[va] if va is not None else None
    B = int(batch_size)
    n_batches = tr.numel() // B                                  # drop_last=True (main.py:294)
    if n_batches < 1:
        raise ValueError(f"batch_size {B} exceeds the training split ({tr.numel()} samples) with drop_last=True")
    pipe = PipelinedStep(store, B, cohort=cohort, n_batches=n_batches, seed=int(seed) * 7919 + 13, epoch_acc=True)
    names = spec.loss_names()
    eval_cache: Dict[int, StepPlan] = {}
    history: List[Dict[str, float]] = []
    best, wait, stopped_epoch, steps = float("inf"), 0, 0, 0

    def write_table():
        """shuffle=True, drop_last=True (main.py:289-298): a fresh device permutation of the training split."""
        perm = tr[torch.randperm(tr.numel(), generator=gen, device=dev)][: n_batches * B]
        if trip:
            pos, neg = sampler.sample(perm, gen)
            table = torch.cat([perm.view(n_batches, B), pos.view(n_batches, B), neg.view(n_batches, B)], dim=1)
            pipe.idx.copy_(table.reshape(-1))
        else:
            pipe.idx.copy_(perm)

    write_table()
    pipe.prime()                      # batch 0 is assembled now; every step assembles the batch of the next one
    epochs_run = 0
    for epoch in range(int(epochs)):
        pipe.epoch_acc.zero_()
        for b in range(n_batches):
            if pipe.epoch_end_next():
                write_table()         # the last step of an epoch prefetches row 0 of the next epoch's table
            if use_graph and pipe.graphs[0] is not None:
                pipe.replay()
            else:
                pipe.step(lr)
                if use_graph:
                    pipe.capture(lr)
            steps += 1
        epochs_run = epoch + 1
        acc = pipe.epoch_acc.detach().cpu().tolist()
        rec = {n: acc[i] / max(acc[-1], 1.0) for i, n in enumerate(names)}
        rec["train_loss"] = acc[len(names)] / max(acc[-1], 1.0)
        if va is not None:
            rec["val_loss"] = _eval_loss(model, store, cohort, va, B, passes, sampler, gen, eval_cache)
        history.append(rec)
        if verbose:
            print(f"[fit] epoch {epoch}: " + ", ".join(f"{k}={v:.5f}" for k, v in rec.items()), flush=True)
        if va is not None and patience and patience > 0:
            # lightning.pytorch.callbacks.EarlyStopping(monitor='val_loss', patience, mode='min', min_delta=0)
            if rec["val_loss"] < best:
                best, wait = rec["val_loss"], 0
            else:
                wait += 1
                if wait >= patience:
                    stopped_epoch = epoch
                    break
        if not np.isfinite(rec["train_loss"]):
            break
    final_val = _eval_loss(model, store, cohort, va, B, passes, sampler, gen, eval_cache) if va is not None else float("nan")
    model._sync_nbt()
    return FitResult(final_val, epochs_run, stopped_epoch, history, steps)


def split_indices(n: int, val_size: float, seed: int):
    """torch.utils.data.random_split semantics (main.py:272-276): num_val = int(n*val_size), seeded permutation."""
    num_val = int(n * val_size)
    g = torch.Generator().manual_seed(int(seed))
    perm = torch.randperm(n, generator=g).tolist()
    return perm[: n - num_val], perm[n - num_val:]


def run_trial(model_class, params: dict, dataset, target_variables, batch_variables=None, surv_event_var=None,
              surv_time_var=None, use_loss_weighting=True, val_size: float = 0.2, early_stop_patience: int = 10,
              seed: int = 0, device=None, use_graph: bool = True):
    """One HPO trial = reference ``objective(params)`` (main.py:228-333): split -> new model -> fit ->
    validate -> (val_loss, epochs, model).  A failed / non-finite trial reports +inf instead of raising so a
    sharded sweep never hangs on a bad configuration."""
    torch.manual_seed(int(seed))
    model = model_class(params, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                        use_loss_weighting, device_type=str(device) if device is not None else None)
    n = len(dataset)
    if model.spec.model == "MultiTripletNetwork":
        lab = np.asarray(dataset.ann[model.main_var])
        n = int((~np.isnan(lab)).sum())
    train_idx, val_idx = split_indices(n, val_size, seed)
    try:
        res = fit(model, dataset, train_idx, val_idx, batch_size=int(params["batch_size"]), epochs=int(params["epochs"]),
                  lr=float(params["lr"]), patience=early_stop_patience, seed=seed, use_graph=use_graph, device=device)
    except (RuntimeError, ValueError) as e:
        return float("inf"), 0, model, {"error": repr(e)}
    val = res.val_loss if np.isfinite(res.val_loss) else float("inf")
    epochs = res.stopped_epoch if res.stopped_epoch else int(params["epochs"])      # main.py:319-322
    return val, epochs, model, {"history": res.history, "steps": res.steps}
