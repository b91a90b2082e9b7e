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

import os

import numpy as np
import torch

from . import ops
from .data import DeviceCohort
from .engine import PipelinedStep, StepPlan


@dataclass
class FitResult:
    val_loss: float
    epochs_run: int
    stopped_epoch: int                      # 0 if early stopping never fired (Lightning's convention)
    history: List[Dict[str, float]] = field(default_factory=list)
    steps: int = 0


_COHORT_LOCK = __import__("threading").Lock()


def _cohort_of(dataset, device) -> DeviceCohort:
    """The dataset's HBM-resident copy, built once and cached on the dataset object.  Fits in flight on several host threads
    share it: it is built under a lock, and PUBLISHED only after the stream that converted / uploaded it has finished -- another
    thread's stream is not ordered behind that one."""
    c = getattr(dataset, "_fx_cohort", None)
    if c is not None and c.device == torch.device(device):
        return c
    with _COHORT_LOCK:
        c = getattr(dataset, "_fx_cohort", None)
        if c is None or c.device != torch.device(device):
            c = DeviceCohort.from_dataset(dataset, device)
            torch.cuda.current_stream(torch.device(device)).synchronize()
            try:
                dataset._fx_cohort = c
            except Exception:
                pass
    return c


# per-epoch validation as a captured hipGraph per chunk size (second use on); FX_EVAL_GRAPHS=0: eager launches (A/B)
EVAL_GRAPHS = __import__("os").environ.get("FX_EVAL_GRAPHS", "1") != "0"


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

    def sample(self, anchors: torch.Tensor, gen):
        """``gen``: an ops.DeviceRng -- one HIP launch (fx_triplet_sample) on the GPU, the training path; or a torch.Generator: the
        same arithmetic stated with torch ops, which is what the host-side semantics tests (and only they) run on CPU tensors."""
        if isinstance(gen, ops.DeviceRng):
            if getattr(self, "_err", None) is None:
                self._err = torch.zeros(1, dtype=torch.int32, device=anchors.device)
            pos, neg = ops.triplet_sample(anchors.contiguous(), self.gid, self.order, self.starts, self.counts, self.rank_in_group,
                                          self.n_groups, gen, self._err)
            if int(self._err.item()):
                self._err.zero_()
                raise ValueError("a label group has a single member: no positive sample exists for its anchor")
            return pos, neg
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


def _eval_loss(model, store, cohort, idx_rows: torch.Tensor, batch_size: int, passes: int, sampler, gen, cache,
               supplied: Optional[dict] = None, epoch: int = 0, use_graph: bool = True) -> float:
    """Mean validation total (batch-size weighted, like Lightning's epoch reduction of validation_step).
    ``supplied`` (parity tests): "val_draws"(epoch, chunk) -> {name: tensor} replaces the in-kernel draws of the VAE family's
    eval forward (z is sampled in eval mode too), "val_triplets"(epoch, chunk) -> (positive rows, negative rows) the device
    triplet sampler."""
    n = idx_rows.numel()
    tot, cnt = 0.0, 0
    acc = []
    vdraws = supplied.get("val_draws") if supplied else None
    vtrip = supplied.get("val_triplets") if supplied else None
    for bi, s in enumerate(range(0, n, batch_size)):
        rows = idx_rows[s:s + batch_size]
        B = rows.numel()
        key = (B, vdraws is not None)
        if key not in cache:
            cache[key] = StepPlan(store, B, train=False, cohort=cohort, n_batches=0, seed=model._seed + 7919 + B,
                                  supplied_draws=vdraws is not None)
        plan = cache[key]
        if passes == 3:
            if vtrip is not None:
                pos, neg = (torch.as_tensor(t, dtype=torch.int64).to(rows.device) for t in vtrip(epoch, bi))
            else:
                pos, neg = sampler.sample(rows, gen)
            plan.idx.copy_(torch.cat([rows, pos, neg]))
        else:
            plan.idx.copy_(rows)
        if vdraws is not None:
            plan.set_draws({k: torch.as_tensor(v).to(rows.device) for k, v in vdraws(epoch, bi).items()})
        plan.eval_step(use_graph=EVAL_GRAPHS and use_graph)      # (fit(use_graph=False) captures nothing at all)
        k = len(plan.spec.loss_names())
        acc.append((plan.loss_vec[k].clone(), B))
    for v, B in acc:
        tot += float(v) * B
        cnt += B
    return tot / max(cnt, 1)


def fit(model, dataset, train_idx: Sequence[int], val_idx: Optional[Sequence[int]] = None, *, batch_size: int,
        epochs: int, lr: float, patience: int = 0, seed: int = 0, use_graph: bool = True, device=None,
        verbose: bool = False, clip: bool = True, frozen: Sequence[str] = (), drop_last: bool = True,
        fresh_optimizer: bool = False, supplied: Optional[dict] = None, prof: Optional[dict] = None,
        precision: Optional[str] = None) -> FitResult:
    """Train ``model`` on ``dataset[train_idx]`` and validate on ``dataset[val_idx]`` once per epoch.
    For MultiTripletNetwork the indices address the valid (non-NaN main label) anchors, like the reference's
    ``TripletMultiOmicDataset`` (data.py:1102-1104).

    Defaults = the HPO trainer (main.py:212-225, :289-298: clip 1.0, shuffle, drop_last=True).  The FineTuner's
    trainer (main.py:530-611) is ``clip=False, drop_last=False, frozen=(...)``: state_dict key prefixes with
    requires_grad=False are neither differentiated nor stepped, and the last partial batch of an epoch is used.

    ``supplied`` (parity tests): {"perms": [per-epoch permutation of range(len(train_idx))], "draws": fn(epoch, batch) ->
    {name: tensor}} replaces the device shuffle and the in-kernel Philox draws by recorded ones; optional "triplets":
    fn(epoch, batch) -> (positive rows, negative rows) replaces the device triplet sampler, "val_draws" / "val_triplets":
    fn(epoch, chunk) the same for the validation batches (epoch == number of epochs run addresses the final validation).
    The schedule (pipelined batch assembly, hipGraph replay, fused kernels) is the production one.
    ``precision``: None = the process default (FX_PRECISION, else "bf16x3", the parity mode); "bf16" = the plain-bf16 throughput mode."""
    if use_graph and ops.in_flight_thread():
        # a fit running beside others on this GPU (trials.run_units(in_flight > 1)) must not capture: a hipGraph capture is
        # process-wide on ROCm and the neighbours synchronise and launch all the time.  Same launches, issued eagerly.
        import warnings
        warnings.warn("fit(use_graph=True) inside trials.run_units(in_flight > 1): launching eagerly instead "
                      "(hipGraph capture is process-wide; pass use_graph=False to silence this)", RuntimeWarning, stacklevel=2)
        use_graph = False
    store = model._bind(device)
    # streams, events and graph capture are keyed on torch's current device: make it the model's for the whole fit
    with torch.cuda.device(store.device):
        return _fit(model, store, dataset, train_idx, val_idx, batch_size=batch_size, epochs=epochs, lr=lr, patience=patience,
                    seed=seed, use_graph=use_graph, verbose=verbose, clip=clip, frozen=frozen, drop_last=drop_last,
                    fresh_optimizer=fresh_optimizer, supplied=supplied, prof=prof, precision=precision)


class _OwnedPlans(dict):
    """The eval-plan cache of one fit: every plan stored in it is registered for close() at the end of the fit."""

    def __init__(self, owned):
        super().__init__()
        self._owned = owned

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        if self._owned is not None:
            self._owned.append(v)


class _Phases:
    """Wall-clock per phase of a fit (``prof`` dict of fit(): diagnostics only; every boundary synchronises the device)."""

    def __init__(self, sink, dev):
        import time
        self.sink, self.dev, self.clock = sink, dev, time.perf_counter
        if sink is not None:
            torch.cuda.synchronize(dev)
            self.t = self.clock()

    def lap(self, name):
        if self.sink is None:
            return
        torch.cuda.synchronize(self.dev)
        now = self.clock()
        self.sink[name] = self.sink.get(name, 0.0) + (now - self.t)
        self.t = now


def _fit(model, store, dataset, train_idx, val_idx, **kw) -> FitResult:
    """fit() proper; every plan it builds is closed when it returns or raises (hipGraph lifetime: ops.retire_graph)."""
    owned: list = []
    try:
        return _fit_impl(model, store, dataset, train_idx, val_idx, owned=owned, **kw)
    finally:
        # the last launches have finished before the plans (and their graphs) are released.  With other fits in flight on this GPU
        # only THIS fit's streams are waited for -- a device-wide synchronisation would stall the neighbours (and is illegal while
        # one of them captures)
        if ops.in_flight_thread():
            torch.cuda.current_stream(store.device).synchronize()
        else:
            torch.cuda.synchronize(store.device)
        for o in owned:
            o.close()


def _fit_impl(model, store, dataset, train_idx, val_idx, *, batch_size, epochs, lr, patience, seed, use_graph, verbose, clip, frozen,
              drop_last, fresh_optimizer, supplied=None, prof=None, owned=None, precision=None) -> FitResult:
    dev = store.device
    ph = _Phases(prof, dev)
    if fresh_optimizer:
        store.reset_optimizer()             # a new torch.optim.Adam per fit (main.py:562-566)
    cohort = _cohort_of(dataset, dev)
    spec = model.spec
    trip = spec.model == "MultiTripletNetwork"
    passes = 3 if trip else 1
    gen = ops.DeviceRng(int(seed))              # Philox stream of this fit's shuffles and triplet draws (HIP kernels, no torch RNG)
    sampler = None
    tr = torch.as_tensor(np.asarray(train_idx), dtype=torch.int64, device=dev)
    va = torch.as_tensor(np.asarray(val_idx), dtype=torch.int64, device=dev) if val_idx is not None and len(val_idx) else None
    if trip:
        sampler = TripletSampler(cohort.ann[model.main_var])
        tr = sampler.valid[tr]
        va = sampler.valid[va] if va is not None else None
    B = int(batch_size)
    n_batches = tr.numel() // B                                  # full batches (drop_last=True: main.py:294)
    tail = 0 if drop_last else tr.numel() - n_batches * B        # DataLoader default keeps the partial batch (main.py:544)
    if n_batches < 1 and tail < 1:
        raise ValueError(f"batch_size {B} exceeds the training split ({tr.numel()} samples) with drop_last=True")
    frozen = tuple(frozen)
    plan_kw = dict(clip=bool(clip), frozen=frozen)
    if precision is not None:           # "bf16": the plain-bf16 throughput mode (training plans; validation plans follow FX_PRECISION)
        plan_kw["precision"] = precision
    if supplied is not None and (tail or (trip and "triplets" not in supplied)):
        raise ValueError("supplied permutations / draws cover full batches only (and the triplet network needs its triplets)")
    ph.lap("setup")
    pipe = PipelinedStep(store, B, cohort=cohort, n_batches=n_batches, seed=int(seed) * 7919 + 13, epoch_acc=True,
                         supplied_draws=supplied is not None, **plan_kw) if n_batches >= 1 else None
    tail_plan = StepPlan(store, tail, train=True, fused=True, supplied_draws=False, seed=int(seed) * 7919 + 17, cohort=cohort,
                         n_batches=0, epoch_acc=True, **plan_kw) if tail else None
    ph.lap("train plans")
    names = spec.loss_names()
    # Validation chunks.  The reference validates in batches of the trial's batch size and Lightning averages the batches weighted by
    # their size (main.py:300-307, :323).  When every head's loss is a plain mean over the rows of a batch -- DirectPred with numerical /
    # categorical heads, no Cox term, and no validation row with a missing label (a masked mean over the VALID rows of a batch weighted
    # by the batch's size is not a global mean) -- that weighted average IS the mean over all validation rows, whatever the chunking
    # (eval-mode BatchNorm and no dropout: rows are independent).  Then the chunks are 128 rows instead of batch_size: a B = 32 trial
    # reads its wide weights 4 x less often per validation pass (13 ms -> 4 ms of a 210 ms trial at the cfg2 shape).  Parity mode
    # (supplied draws) and every other model keep the reference's chunks.
    Bv = B
    if (va is not None and supplied is None and spec.model == "DirectPred" and spec.surv_event_var is None and B < 128
            and os.environ.get("FX_VAL_CHUNK", "1") != "0"):
        ok = True
        for (v, kind, C) in spec.variables:
            lab = cohort.ann[v][va]
            ok = ok and not bool(torch.isnan(lab).any()) and (kind == "numerical" or bool(((lab >= 0) & (lab < C)).all()))
        if ok:
            Bv = 128
    eval_cache: Dict[int, StepPlan] = _OwnedPlans(owned)
    for o in (pipe, tail_plan):
        if o is not None:
            owned.append(o)
    history: List[Dict[str, float]] = []
    best, wait, stopped_epoch, steps = float("inf"), 0, 0, 0
    tails: List[torch.Tensor] = []          # tail rows of the epochs whose table has been drawn, oldest first

    def rows_of(perm, k, epoch=None):
        if not trip:
            return perm
        if supplied is not None and epoch is not None:      # recorded positives / negatives, batch by batch
            pn = [supplied["triplets"](epoch, b) for b in range(perm.numel() // k)]
            pos = torch.cat([torch.as_tensor(p_, dtype=torch.int64) for p_, _ in pn]).to(dev)
            neg = torch.cat([torch.as_tensor(n_, dtype=torch.int64) for _, n_ in pn]).to(dev)
        else:
            pos, neg = sampler.sample(perm, gen)
        return torch.cat([perm.view(-1, k), pos.view(-1, k), neg.view(-1, k)], dim=1).reshape(-1)

    tables_written = [0]

    def write_table():
        """shuffle=True (main.py:289-298): a fresh device permutation of the training split."""
        if supplied is not None:
            e = min(tables_written[0], len(supplied["perms"]) - 1)      # (the table after the last epoch is never trained on)
            perm = tr[torch.as_tensor(supplied["perms"][e], dtype=torch.int64).to(dev)]
        else:
            perm = ops.randperm(tr.numel(), gen, dev, src=tr)
        e_written = tables_written[0]
        tables_written[0] += 1
        if pipe is not None:
            pipe.idx.copy_(rows_of(perm[: n_batches * B], B, e if supplied is not None and e_written < len(supplied["perms"]) else None))
        if tail:
            tails.append(rows_of(perm[n_batches * B:], tail))

    write_table()
    if pipe is not None:
        pipe.prime()                  # batch 0 is assembled now; every step assembles the batch of the next one
    ph.lap("prime")
    epochs_run = 0
    for epoch in range(int(epochs)):
        if pipe is not None:
            pipe.epoch_acc.zero_()
        if tail_plan is not None:
            tail_plan.epoch_acc.zero_()
        for b in range(n_batches):
            if pipe.epoch_end_next():
                write_table()         # the last step of an epoch prefetches row 0 of the next epoch's table
            if supplied is not None:
                pipe.pending.set_draws({k: torch.as_tensor(v).to(dev) for k, v in supplied["draws"](epoch, b).items()})
            if use_graph and pipe.graphs[0] is not None:
                pipe.replay()
            else:
                pipe.step(lr)
                ph.lap("first step (eager)")
                if use_graph:
                    pipe.capture(lr)
                    ph.lap("graph capture")
            steps += 1
        if tail_plan is not None:
            if pipe is None:
                write_table()
            tail_plan.idx.copy_(tails.pop(0))
            tail_plan.train_step(lr, gather=True)
            if pipe is not None:
                pipe.refresh()        # the pending batch's fused wide forward was computed before this step changed the weights
            steps += 1
        epochs_run = epoch + 1
        ph.lap("steps")
        # epoch means weighted by batch size, like Lightning's on_epoch reduction of the logged losses
        acc = [0.0] * (len(names) + 1)
        wsum = 0.0
        for pl, bs in ((pipe, B), (tail_plan, tail)):
            if pl is None:
                continue
            a = pl.epoch_acc.detach().cpu().tolist()
            for i in range(len(names) + 1):
                acc[i] += a[i] * bs
            wsum += a[-1] * bs
        rec = {n: acc[i] / max(wsum, 1.0) for i, n in enumerate(names)}
        rec["train_loss"] = acc[len(names)] / max(wsum, 1.0)
        if va is not None:
            rec["val_loss"] = _eval_loss(model, store, cohort, va, Bv, passes, sampler, gen, eval_cache, supplied, epoch, use_graph)
        ph.lap("epoch readback + validation")
        history.append(rec)
        if verbose:
            print(f"[fit] epoch {epoch}: " + ", ".join(f"{k}={v:.5f}" for k, v in rec.items()), flush=True)
        if va is not None and patience and patience > 0:
            # lightning.pytorch.callbacks.EarlyStopping(monitor='val_loss', patience, mode='min', min_delta=0)
            if not np.isfinite(rec["val_loss"]):          # EarlyStopping(check_finite=True): a non-finite monitor stops at once
                stopped_epoch = epoch
                break
            if rec["val_loss"] < best:
                best, wait = rec["val_loss"], 0
            else:
                wait += 1
                if wait >= patience:
                    stopped_epoch = epoch
                    break
        if not np.isfinite(rec["train_loss"]):
            break
    final_val = (_eval_loss(model, store, cohort, va, Bv, passes, sampler, gen, eval_cache, supplied, epochs_run, use_graph)
                 if va is not None else float("nan"))
    ph.lap("final validation")
    model._sync_nbt()
    return FitResult(final_val, epochs_run, stopped_epoch, history, steps)


def kfold_indices(n: int, n_splits: int, seed: int):
    """sklearn.model_selection.KFold(n_splits, shuffle=True) fold sizes and semantics (reference main.py:505, :517):
    the first n % n_splits folds get one extra sample; a seeded permutation replaces sklearn's unseeded one."""
    g = torch.Generator().manual_seed(int(seed))
    perm = torch.randperm(n, generator=g).tolist()
    sizes = [n // n_splits + (1 if i < n % n_splits else 0) for i in range(n_splits)]
    folds, o = [], 0
    for sz in sizes:
        val = sorted(perm[o:o + sz])
        held = set(val)
        folds.append(([i for i in range(n) if i not in held], val))
        o += sz
    return folds


FREEZE_PREFIXES = {"encoders": "encoders.", "supervisors": "MLPs."}       # apply_freeze_config, main.py:530-539


def _fine_tune_units(model, lrs, cfgs, n_splits):
    return [(lr, cfg, fi) for lr in lrs for cfg in cfgs for fi in range(n_splits)]


def fine_tune(model, dataset, *, n_splits: int = 5, batch_size: int = 32, learning_rates=None, max_epoch: int = 50,
              freeze_configs=None, seed: int = 0, device=None, use_graph: bool = True, verbose: bool = False,
              sharded: bool = False, schedule: str = "queue", comm_device=None, supplied_for=None, details: Optional[dict] = None):
    """The reference's ``FineTuner.run_experiments`` (main.py:575-659) on the engine: for every learning rate x
    freeze configuration, k-fold cross-validated short fits of a deep copy of ``model`` (fresh Adam, no gradient
    clipping, partial last batch kept, early stopping with patience 3), pick the configuration with the lowest mean
    validation loss, then continue training on all samples for the mean stopped epoch of that configuration.
    Frozen groups cost nothing: their backward, norm and Adam work is not launched.

    ``sharded=True`` (inside an initialised torch.distributed job, one process per GPU): the lr x freeze x fold fits
    -- 45 with the defaults, all starting from the same weights and independent of each other -- are the units of
    ``trials.run_units`` (claimed longest-first from a shared counter); one all_gather collects (val_loss, stopped
    epoch); the rank that ran the LAST fit, from which the reference continues (main.py:647), trains the final model
    and broadcasts its weights.  Rank 0's starting weights are broadcast first, so every rank fine-tunes the same model.

    ``supplied_for(unit)`` (parity tests; unit = (lr index, configuration index, fold) or "final") returns the ``supplied`` dict of that
    fit (recorded shuffles / dropout draws, see fit()); ``details``, if given, receives {unit: (val_loss, stopped_epoch)}.

    Returns (final_model, best, results) with ``results`` = the reference's ``val_loss_results`` records."""
    import copy
    lrs = list(learning_rates) if learning_rates else [model.config["lr"], model.config["lr"] / 10, model.config["lr"] / 100]
    cfgs = list(freeze_configs) if freeze_configs else [{"encoders": True, "supervisors": False},
                                                         {"encoders": False, "supervisors": True},
                                                         {"encoders": False, "supervisors": False}]
    n = len(dataset)
    if model.spec.model == "MultiTripletNetwork":          # the FineTuner wraps the dataset in TripletMultiOmicDataset
        n = int((~np.isnan(np.asarray(dataset.ann[model.main_var]))).sum())
    folds = kfold_indices(n, n_splits, seed)

    def frozen_of(cfg):
        return tuple(FREEZE_PREFIXES[k] for k in ("encoders", "supervisors") if cfg.get(k))

    def unit_key(lr, cfg, fi):
        return (lrs.index(lr), cfgs.index(cfg), fi)

    def one_fit(lr, cfg, fi):
        m = copy.deepcopy(model)
        tr, va = folds[fi]
        res = fit(m, dataset, tr, va, batch_size=batch_size, epochs=max_epoch, lr=float(lr), patience=3,
                  seed=seed * 1000 + fi, use_graph=use_graph, device=device, clip=False, frozen=frozen_of(cfg),
                  drop_last=False, fresh_optimizer=True,
                  supplied=supplied_for(unit_key(lr, cfg, fi)) if supplied_for is not None else None)
        if details is not None:
            details[unit_key(lr, cfg, fi)] = (float(res.val_loss), int(res.stopped_epoch))
        return m, res

    def final_fit(last, best):
        # main.py:647-659: the final model continues from the LAST cross-validation model, on all samples
        final = copy.deepcopy(last)
        if best["epochs"] > 0:
            fit(final, dataset, list(range(n)), None, batch_size=batch_size, epochs=best["epochs"], lr=float(best["learning_rate"]),
                seed=seed * 1000 + 999, use_graph=use_graph, device=device, clip=False, frozen=frozen_of(best["freeze"]),
                drop_last=False, fresh_optimizer=True, supplied=supplied_for("final") if supplied_for is not None else None)
        return final

    def summarise(vals, eps):
        results, o = [], 0
        for lr in lrs:
            for cfg in cfgs:
                v, e = vals[o:o + n_splits], eps[o:o + n_splits]
                o += n_splits
                rec = {"learning_rate": lr, "average_val_loss": float(np.mean(v)), "freeze": cfg, "epochs": int(np.mean(e))}
                results.append(rec)
                if verbose:
                    print(f"[fine_tune] lr {lr} freeze {cfg}: val_loss {rec['average_val_loss']:.5f}, epochs {rec['epochs']}", flush=True)
        return results, min(results, key=lambda r: r["average_val_loss"])

    units = _fine_tune_units(model, lrs, cfgs, n_splits)
    if not sharded:
        vals, eps, last = [], [], model
        for (lr, cfg, fi) in units:
            last, res = one_fit(lr, cfg, fi)
            vals.append(res.val_loss)
            eps.append(res.stopped_epoch)
        results, best = summarise(vals, eps)
        return final_fit(last, best), best, results

    import torch.distributed as dist
    from . import trials
    from .models.base import resolve_device
    # buffers of the collectives live on the model's GPU (RCCL); ``comm_device`` overrides that for the gloo protocol tests
    dev = torch.device(comm_device) if comm_device is not None else resolve_device(device)
    shapes = model.spec.state_shapes()
    rank = dist.get_rank() if dist.is_initialized() else 0
    # identical starting weights on every rank
    model.load_state_dict(trials.broadcast_state(model.state_dict() if rank == 0 else None, shapes, 0, dev))
    last_uid = len(units) - 1
    n_tr, n_va = len(folds[0][0]), len(folds[0][1])
    P = float(sum(int(np.prod(shp)) for k, shp in shapes.items() if len(shp) == 2 and int(np.prod(shp)) >= (1 << 20)
                  and k.startswith("encoders.")))        # the encoders' wide weights: what a frozen-encoder fit does not stream

    def unit_cost(u):
        lr, cfg, fi = u
        frozen_enc = bool(cfg.get("encoders"))
        step = trials.COST_STEP_FIXED_S + (0.0 if frozen_enc else trials.COST_STEP_PER_PARAM_S * P) + trials.COST_VAL_PER_PARAM_S * P
        return trials.COST_FIT_FIXED_S + max_epoch * (-(-n_tr // batch_size)) * step

    def unit_fn(uid):
        lr, cfg, fi = units[uid]
        m, res = one_fit(lr, cfg, fi)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()} if uid == last_uid else None
        # the early-stopping epoch rides in the "epochs" column; a fit that never stopped early reports 0 (Lightning)
        return res.val_loss, res.stopped_epoch, sd

    table, held = trials.run_units(len(units), unit_fn, [unit_cost(u) for u in units], dev, keep=[last_uid], schedule=schedule)
    results, best = summarise(table[:, 1].tolist(), table[:, 2].tolist())
    owner = int(table[last_uid, 4]) if (table[last_uid, 4] == table[last_uid, 4] and table[last_uid, 4] >= 0) else 0
    final_sd = None
    if rank == owner and last_uid in held:
        last = copy.deepcopy(model)
        last.load_state_dict(held[last_uid])
        final_sd = final_fit(last, best).state_dict()
    held_final = {last_uid: final_sd} if final_sd is not None else {}
    state = trials.agree_and_broadcast_state(held_final, last_uid, table, shapes, dev)
    final = copy.deepcopy(model)
    if state is not None:
        final.load_state_dict(state)
    return final, best, results


def split_indices(n: int, val_size: float, seed: int):
    """torch.utils.data.random_split semantics (main.py:272-276): num_val = int(n*val_size), seeded permutation."""
    num_val = int(n * val_size)
    g = torch.Generator().manual_seed(int(seed))
    perm = torch.randperm(n, generator=g).tolist()
    return perm[: n - num_val], perm[n - num_val:]


def _new_model(model_class, params, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
               use_loss_weighting, device, model_kwargs):
    # keyword arguments, like the reference's model_args dict (main.py:230-261): CrossModalPred's positional order
    # differs (input_layers / output_layers come before use_loss_weighting) and it takes them through **model_kwargs
    return model_class(config=params, dataset=dataset, target_variables=target_variables, batch_variables=batch_variables,
                       surv_event_var=surv_event_var, surv_time_var=surv_time_var, use_loss_weighting=use_loss_weighting,
                       device_type=str(device) if device is not None else None, **model_kwargs)


def _n_loader_samples(model, dataset) -> int:
    """len(loader_dataset): the triplet network trains on the valid anchors only (main.py:176-181, data.py:1102-1104)."""
    if model.spec.model == "MultiTripletNetwork":
        return int((~np.isnan(np.asarray(dataset.ann[model.main_var]))).sum())
    return len(dataset)


def trial_splits(n: int, val_size: float, seed: int, use_cv: bool = False, n_splits: int = 5):
    """The split iterator of ``objective`` (main.py:266-281): KFold(n_splits, shuffle=True) folds, or the single
    random_split in the same (train, val) format."""
    return kfold_indices(n, n_splits, seed) if use_cv else [split_indices(n, val_size, seed)]


def _short_fit_placement(fn):
    """HPO trials are short fits (a few dozen steps) that do not earn back a per-weight search over placements (each rejected
    candidate costs ~6 ms and a trip of its blocks back to the driver).  Their wide weights come from the process's partition arena
    like everyone's (engine.PartitionArena: W and m, v on two sides of a memory-partition boundary, nothing to search); where there is
    no arena, from the process-level pool when an earlier model of the same shape left rated arrays (the folds of a cross-validated
    trial), else the first placement (engine.placement_tries(1); FX_PLACEMENT_TRIES_TRIAL=<n> searches, FX_PLACEMENT_TRIES overrides
    everything)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        from .engine import placement_tries
        with placement_tries(int(os.environ.get("FX_PLACEMENT_TRIES_TRIAL", "1"))):
            return fn(*a, **kw)
    return wrapped


@_short_fit_placement
def run_trial_fold(model_class, params: dict, dataset, target_variables, train_idx, val_idx, batch_variables=None,
                   surv_event_var=None, surv_time_var=None, use_loss_weighting=True, early_stop_patience: int = 10,
                   seed: int = 0, device=None, use_graph: bool = True, **model_kwargs):
    """One pass of ``objective``'s loop body (main.py:283-326): new model -> fit with early stopping -> validate.
    Returns (val_loss, epochs, model, info); a failed / non-finite fit reports +inf instead of raising so a sharded
    sweep never hangs on a bad configuration.  This is the unit that cross-validated sweeps shard over the GPUs."""
    with _CTOR_LOCK:          # seed + constructor draw from the process-wide generators: one at a time when trials run on threads
        torch.manual_seed(int(seed))
        model = _new_model(model_class, params, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                           use_loss_weighting, device, model_kwargs)
    try:
        res = fit(model, dataset, train_idx, val_idx, batch_size=int(params["batch_size"]), epochs=int(params["epochs"]),
                  lr=float(params["lr"]), patience=early_stop_patience, seed=seed, use_graph=use_graph, device=device)
    except (RuntimeError, ValueError) as e:
        return float("inf"), 0, model, {"error": repr(e)}
    val = res.val_loss if np.isfinite(res.val_loss) else float("inf")
    epochs = res.stopped_epoch if res.stopped_epoch else int(params["epochs"])      # main.py:319-322
    return val, epochs, model, {"history": res.history, "steps": res.steps}


_CTOR_LOCK = __import__("threading").Lock()
FOLD_SEED_STRIDE = 7919      # fold i of a trial seeds its model / shuffles with seed + i * stride (fold 0 = the single split's seed)


@_short_fit_placement
def run_trial(model_class, params: dict, dataset, target_variables, batch_variables=None, surv_event_var=None,
              surv_time_var=None, use_loss_weighting=True, val_size: float = 0.2, early_stop_patience: int = 10,
              seed: int = 0, device=None, use_graph: bool = True, use_cv: bool = False, n_splits: int = 5, **model_kwargs):
    """One HPO trial = reference ``objective(params)`` (main.py:228-333): split(s) -> per split a new model -> fit ->
    validate -> (mean val_loss, int(mean epochs), the last model trained).  ``use_cv`` selects the k-fold branch
    (main.py:267-269); the mean over folds is what the optimiser is told (:327-333).  With cross-validation the caller
    rebuilds the final model on all samples (``full_train``, main.py:403-414)."""
    # len(loader_dataset): the triplet network trains on the valid anchors only (main.py:176-181, data.py:1102-1104)
    n = len(dataset)
    if getattr(model_class, "__name__", "") == "MultiTripletNetwork":
        n = int((~np.isnan(np.asarray(dataset.ann[target_variables[0]]))).sum())
    vals, eps, infos, model = [], [], [], None
    for fi, (tr, va) in enumerate(trial_splits(n, val_size, seed, use_cv, n_splits)):
        val, ep, model, info = run_trial_fold(model_class, params, dataset, target_variables, tr, va, batch_variables,
                                              surv_event_var, surv_time_var, use_loss_weighting, early_stop_patience,
                                              int(seed) + fi * FOLD_SEED_STRIDE, device, use_graph, **model_kwargs)
        if "error" in info:
            return float("inf"), 0, model, info
        vals.append(val)
        eps.append(ep)
        infos.append(info)
    if not use_cv:
        return vals[0], eps[0], model, infos[0]
    mean_val = float(np.mean(vals))
    info = {"fold_val_losses": vals, "fold_epochs": eps, "steps": sum(i.get("steps", 0) for i in infos),
            "history": [i.get("history") for i in infos]}
    return (mean_val if np.isfinite(mean_val) else float("inf")), int(np.mean(eps)), model, info


def full_train(model_class, params: dict, dataset, target_variables, batch_variables=None, surv_event_var=None,
               surv_time_var=None, use_loss_weighting=True, seed: int = 0, device=None, use_graph: bool = True,
               **model_kwargs):
    """``objective(params, full_train=True)`` (main.py:246-262): a new model trained on ALL samples for params["epochs"]
    epochs -- shuffle, drop_last, clip 1.0, no validation, no early stopping -- the final model of a cross-validated
    search (main.py:403-414, where ``epochs`` is the best trial's mean stopped epoch)."""
    torch.manual_seed(int(seed))
    model = _new_model(model_class, params, dataset, target_variables, batch_variables, surv_event_var, surv_time_var,
                       use_loss_weighting, device, model_kwargs)
    n = _n_loader_samples(model, dataset)
    res = fit(model, dataset, list(range(n)), None, batch_size=int(params["batch_size"]), epochs=int(params["epochs"]),
              lr=float(params["lr"]), patience=0, seed=seed, use_graph=use_graph, device=device)
    return model, {"history": res.history, "steps": res.steps}
