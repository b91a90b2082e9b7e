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
import time

import numpy as np
import torch
import torch.distributed as dist

from . import trials
from .arch import spec_from_dataset
from .data import MultiOmicDataset
from .fit import run_trial
from .models import DirectPred


def _cohort(layers, n, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dat = {name: torch.randn(n, F, generator=g, device=device) for name, F in layers}
    first = dat[layers[0][0]]
    ann = {"y": first[:, :16].sum(1) / 4 + 0.1 * torch.randn(n, generator=g, device=device)}
    return dat, ann


def run_cfg5(dev, n_trials: int = 64, epochs: int = 3, features: int = 20000, samples: int = 2048, seed: int = 0,
             keep_winner: bool = True) -> dict:
    """The cfg5 workload on the CURRENT process group (or a single process): rank 0 builds the synthetic cohort and
    broadcasts it, trials are assigned longest-first, every rank runs its trials with the engine loop, one all_gather
    collects the records and the winner's state_dict is broadcast.  Returns the summary dict (identical on every rank)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    layers = [("gex", features), ("cnv", features)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dat, ann = _cohort(layers, samples, dev, 1234) if rank == 0 else (None, None)
    torch.cuda.synchronize(dev)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    dat, ann = trials.broadcast_cohort(dat, ann, dev)
    torch.cuda.synchronize(dev)
    t_bcast = time.perf_counter() - t0
    feats = {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(samples)], {})
    plist = trials.draw_search_space(n_trials, seed=seed, epochs=epochs)
    n_train = samples - int(samples * 0.2)
    costs = [trials.trial_cost(p, 2 * features, n_train) for p in plist]
    stats = {"samples": 0, "busy": 0.0}

    def trial_fn(tid, params):
        t = time.perf_counter()
        val, ep, model, info = run_trial(DirectPred, params, ds, ["y"], early_stop_patience=0, seed=seed * 100003 + tid,
                                         device=dev)
        if "error" in info:
            raise RuntimeError(info["error"])
        stats["samples"] += info["steps"] * int(params["batch_size"])
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()} if keep_winner else None
        del model
        torch.cuda.synchronize(dev)
        stats["busy"] += time.perf_counter() - t
        return val, ep, sd

    def shapes_of(params):
        return spec_from_dataset("DirectPred", params, ds, ["y"]).state_shapes()

    t1 = time.perf_counter()
    table, best, state = trials.run_sweep(plist, trial_fn, costs, dev, shapes_of if keep_winner else None)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t1
    mine = torch.tensor([float(stats["samples"]), wall, stats["busy"]], dtype=torch.float64, device=dev)
    if world > 1:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        allr = torch.stack(parts).cpu().numpy()
    else:
        allr = mine.cpu().numpy()[None]
    ok = table[:, 3] == trials.STATUS_OK
    busy = allr[:, 2]
    return {
        "workload": f"cfg5: {n_trials} DirectPred trials (2 x {features} features, N={samples}, {epochs} epochs), "
                    f"{world} GPU(s), trial sharding (LPT)",
        "n_gpus": world, "trials": int(n_trials), "trials_ok": int(ok.sum()), "best_trial": best,
        "best_val_loss": float(table[best, 1]), "best_params": plist[best],
        "winner_state_tensors": len(state) if state is not None else 0,
        "cohort_generate_s": round(t_gen, 4), "cohort_broadcast_s": round(t_bcast, 4),
        "sweep_wall_s": round(float(allr[:, 1].max()), 3),
        "aggregate_samples_per_s": round(float(allr[:, 0].sum()) / float(allr[:, 1].max()), 1),
        "rank_busy_s": [round(float(b), 3) for b in busy],
        "tail_imbalance": round(1.0 - float(busy.mean()) / max(float(busy.max()), 1e-9), 4),
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--features", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    out = run_cfg5(dev, a.trials, a.epochs, a.features, a.samples, a.seed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
