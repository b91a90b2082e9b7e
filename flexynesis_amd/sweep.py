"""BASELINE configs[4] (cfg5): a DirectPred HPO sweep sharded over the GPUs of one node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
           -m flexynesis_amd.sweep --trials 64 --epochs 3

Rank 0 builds the (synthetic) cohort and broadcasts it once over RCCL; the trial list is drawn
from the reference's search space (config.py:7-15) with a fixed seed on every rank; trials are assigned
longest-first; every rank runs its trials with the engine loop (reference ``objective``, main.py:228-333, early
stopping off so the work per trial is deterministic); one all_gather collects (val_loss, epochs) and the winner's
state_dict is broadcast.  Rank 0 prints one JSON line with the result table summary and the aggregate throughput.
"""
from __future__ import annotations

import argparse
import json
import os
import threading
import time
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import trials
from .arch import spec_from_dataset
from .data import MultiOmicDataset
from .fit import FOLD_SEED_STRIDE, full_train, run_trial, run_trial_fold, trial_splits
from .models import DirectPred


def _cohort(layers, n, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dat = {name: torch.randn(n, F, generator=g, device=device) for name, F in layers}
    first = dat[layers[0][0]]
    ann = {"y": first[:, :16].sum(1) / 4 + 0.1 * torch.randn(n, generator=g, device=device)}
    return dat, ann


def run_cfg5(dev, n_trials: int = 64, epochs: int = 3, features: int = 20000, samples: int = 2048, seed: int = 0,
             keep_winner: bool = True, schedule: str = "queue", force_collectives: bool = False, use_cv: bool = False,
             n_splits: int = 5, in_flight: Optional[int] = None, use_graph: Optional[bool] = None) -> dict:
    """The cfg5 workload on the CURRENT process group (or a single process): rank 0 builds the synthetic cohort and
    broadcasts it, the trials are claimed longest-first from a counter shared by the ranks (``schedule="static"``: the
    LPT assignment computed up front), every rank runs its trials with the engine loop, one all_gather collects the
    records and the winner's state_dict is broadcast.  ``use_cv``: the reference's cross-validated search
    (main.py:267-269, :403-414) -- the unit of sharding is then one (trial, fold) fit, the optimiser's figure of merit the
    mean over a trial's folds, and the final model is rebuilt on all samples by rank 0 and broadcast.
    ``in_flight`` trials run concurrently on every GPU (host threads with their own streams, eager launches:
    trials.run_units); ``use_graph`` (default: only with one trial in flight) replays hipGraphs inside a trial.
    Returns the summary dict (identical on every rank)."""
    world0 = dist.get_world_size() if dist.is_initialized() else 1
    if in_flight is None:
        # four trials in flight per GPU when a rank has at least two rounds of them (8 trials per GPU: 27.0-29.2 k samples/s with two in
        # flight, 26.9-29.4 k with three, 28.4-30.5 k with four, scripts/sweep_inflight_ab.py), fewer for short lists (tail imbalance)
        in_flight = max(1, min(4, (int(n_trials) * (n_splits if use_cv else 1)) // (2 * world0)))
    use_graph = (int(in_flight) <= 1) if use_graph is None else bool(use_graph)
    if int(in_flight) > 1 and use_graph:
        raise ValueError("trials in flight on several threads must not capture hipGraphs: use_graph=False")
    slock = threading.Lock()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    layers = [("gex", features), ("cnv", features)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    # (a rank that fails BEFORE a collective leaves the others waiting in it: rank 0 announces whether it has a cohort first, and
    # every rank returns the same error record instead of entering the broadcast)
    gen_err = None
    try:
        dat, ann = _cohort(layers, samples, dev, 1234) if rank == 0 else (None, None)
    except Exception as e:
        dat, ann, gen_err = None, None, repr(e)
    if world > 1 or (force_collectives and dist.is_initialized()):
        msg = [gen_err]
        dist.broadcast_object_list(msg, src=0)
        gen_err = msg[0]
    if gen_err is not None:
        return {"error": f"rank 0 could not build the cohort: {gen_err}", "n_gpus": world, "trials": int(n_trials), "trials_ok": 0,
                "trial_val_losses": [float("inf")] * int(n_trials), "aggregate_samples_per_s": 0.0}
    torch.cuda.synchronize(dev)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    dat, ann = trials.broadcast_cohort(dat, ann, dev, force_collectives=force_collectives)
    torch.cuda.synchronize(dev)
    t_bcast = time.perf_counter() - t0
    feats = {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(samples)], {})
    plist = trials.draw_search_space(n_trials, seed=seed, epochs=epochs)
    stats = {"samples": 0, "busy": 0.0}
    def shapes_of(params):
        return spec_from_dataset("DirectPred", params, ds, ["y"]).state_shapes()

    def account(t, info, params):
        if "error" in info:
            raise RuntimeError(info["error"])
        torch.cuda.current_stream(dev).synchronize()
        with slock:
            stats["samples"] += info["steps"] * int(params["batch_size"])
            stats["busy"] += time.perf_counter() - t

    # device memory of the largest trial: 12 B per parameter (weights + Adam moments) and the split operands / activations on top
    specs = [spec_from_dataset("DirectPred", p, ds, ["y"]) for p in plist]
    unit_bytes = 1.25 * 12.0 * max(sp.param_count() for sp in specs) + (1 << 30)
    phases: dict = {}
    t1 = time.perf_counter()
    if not use_cv:
        n_val = int(samples * 0.2)
        costs = [trials.trial_cost(p, [features, features], samples - n_val, n_val) for p in plist]

        def trial_fn(tid, params):
            t = time.perf_counter()
            val, ep, model, info = run_trial(DirectPred, params, ds, ["y"], early_stop_patience=0, seed=seed * 100003 + tid,
                                             device=dev, use_graph=use_graph)
            # only a trial that beats this rank's best so far can be the winner: the others' weights are never cloned
            sd = None
            with slock:
                better = keep_winner and "error" not in info and val <= stats.get("best", float("inf"))
                if better:
                    stats["best"] = val
            if better:
                sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
            del model
            account(t, info, params)
            return val, ep, sd

        table, best, state = trials.run_sweep(plist, trial_fn, costs, dev, shapes_of if keep_winner else None,
                                              schedule=schedule, force_collectives=force_collectives, in_flight=in_flight,
                                              unit_bytes=unit_bytes, timings=phases)
        trial_vals = table[:, 1]
        ok = table[:, 3] == trials.STATUS_OK
    else:
        splits = trial_splits(samples, 0.2, seed, True, n_splits)
        units = [(tid, fi) for tid in range(n_trials) for fi in range(n_splits)]
        costs = [trials.trial_cost(plist[tid], [features, features], len(splits[fi][0]), len(splits[fi][1])) for tid, fi in units]

        def unit_fn(uid):
            tid, fi = units[uid]
            t = time.perf_counter()
            val, ep, model, info = run_trial_fold(DirectPred, plist[tid], ds, ["y"], splits[fi][0], splits[fi][1],
                                                  early_stop_patience=0, seed=seed * 100003 + tid + fi * FOLD_SEED_STRIDE, device=dev,
                                                  use_graph=use_graph)
            del model
            account(t, info, plist[tid])
            return val, ep, None

        utable, _ = trials.run_units(len(units), unit_fn, costs, dev, keep=[], schedule=schedule, force_collectives=force_collectives,
                                     in_flight=in_flight, unit_bytes=unit_bytes, timings=phases)
        per_trial = utable[:, 1].reshape(n_trials, n_splits)
        trial_vals = per_trial.mean(axis=1)                       # main.py:327-333: the mean over the folds
        trial_eps = utable[:, 2].reshape(n_trials, n_splits).mean(axis=1).astype(int)
        ok = (utable[:, 3].reshape(n_trials, n_splits) == trials.STATUS_OK).all(axis=1)
        best = int(np.argmin(np.where(ok, trial_vals, np.inf)))
        state = None
        if keep_winner and np.isfinite(trial_vals[best]):
            # main.py:403-414: the final model is rebuilt on ALL samples with the best parameters and its mean epochs
            held = {}
            if rank == 0:
                t = time.perf_counter()
                final, info = full_train(DirectPred, dict(plist[best], epochs=max(int(trial_eps[best]), 1)), ds, ["y"],
                                         seed=seed * 100003 + 99991, device=dev, use_graph=use_graph)
                held[0] = {k: v.detach() for k, v in final.state_dict().items()}
                account(t, info, plist[best])
            owner_table = np.zeros((1, 5))
            state = trials.agree_and_broadcast_state(held, 0, owner_table, shapes_of(plist[best]), dev, force_collectives)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t1
    mine = torch.tensor([float(stats["samples"]), wall, stats["busy"]], dtype=torch.float64, device=trials.collective_device(dev))
    if world > 1 or (force_collectives and dist.is_initialized()):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        allr = torch.stack(parts).cpu().numpy()
    else:
        allr = mine.cpu().numpy()[None]
    busy = allr[:, 2]
    # What the trial list allows at best on this many GPUs: every optimisation step streams the trial's wide weights and Adam moments
    # once (24 B per parameter) and every validation chunk its weights (4 B), at the rate the dominant kernel reaches on a fast
    # placement (5.4 TB/s, bench.py roofline.achieved), nothing else costs anything, the ranks are perfectly balanced.
    n_tr, n_va = samples - int(samples * 0.2), int(samples * 0.2)
    stream_s, stream_samples = 0.0, 0
    for sp, p_ in zip(specs, plist):
        bsz = int(p_["batch_size"])
        wide = float(sum(int(np.prod(shp)) for k, shp in sp.state_shapes().items() if len(shp) == 2 and int(np.prod(shp)) >= (1 << 20)))
        steps = int(p_["epochs"]) * (n_tr // bsz) * (n_splits if use_cv else 1)
        vals = (int(p_["epochs"]) + 1) * (-(-n_va // 64)) * (n_splits if use_cv else 1)
        stream_s += (steps * 24.0 + vals * 4.0) * wide / 5.4e12
        stream_samples += steps * bsz
    return {
        "workload": f"cfg5: {n_trials} DirectPred trials (2 x {features} features, N={samples}, {epochs} epochs"
                    + (f", {n_splits}-fold CV + final model on all samples" if use_cv else "") + f"), "
                    f"{world} GPU(s), trial sharding ({'work queue, longest first' if schedule == 'queue' and world > 1 else 'LPT'})",
        "n_gpus": world, "trials_in_flight_per_gpu": int(in_flight), "hipgraph_replay": bool(use_graph), "trials": int(n_trials), "trials_ok": int(ok.sum()), "best_trial": best,
        "best_val_loss": float(trial_vals[best]), "best_params": plist[best],
        "trial_val_losses": [float(v) for v in trial_vals],
        "winner_state_tensors": len(state) if state is not None else 0,
        "cohort_generate_s": round(t_gen, 4), "cohort_broadcast_s": round(t_bcast, 4),
        "rank0_phases_s": phases, "failed_units_rank0": dict(trials.LAST_ERRORS),
        "sweep_wall_s": round(float(allr[:, 1].max()), 3),
        "aggregate_samples_per_s": round(float(allr[:, 0].sum()) / float(allr[:, 1].max()), 1),
        "stream_bound_samples_per_s": round(world * stream_samples / max(stream_s, 1e-9), 1),
        "stream_bound_note": "trial list's steps x batch over the time to stream every step's wide weights + Adam moments (24 B/param; validation "
                             "4 B/param) at 5.4 TB/s, no fixed cost per step or trial, perfect balance over the ranks",
        "rank_busy_s": [round(float(b), 3) for b in busy],
        "busy_over_wall": round(float(busy.sum()) / (world * max(int(in_flight), 1) * max(float(allr[:, 1].max()), 1e-9)), 4),
        "tail_imbalance": round(1.0 - float(busy.mean()) / max(float(busy.max()), 1e-9), 4),
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--features", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cv", type=int, default=0, help="k > 1: k-fold cross-validated trials (units = trial x fold)")
    ap.add_argument("--schedule", default="queue", choices=["queue", "static"])
    ap.add_argument("--in-flight", type=int, default=0, help="trials running concurrently per GPU (host threads, eager launches); 0 = up to four")
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("FX_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("FX_BENCH_BACKEND", "nccl") == "gloo":       # collectives through host memory (ranks may share a GPU)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    out = run_cfg5(dev, a.trials, a.epochs, a.features, a.samples, a.seed, schedule=a.schedule, use_cv=a.cv > 1,
                   n_splits=max(a.cv, 2), in_flight=a.in_flight or None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
