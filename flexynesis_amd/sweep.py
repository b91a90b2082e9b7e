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
    layers = [("gex", a.features), ("cnv", a.features)]
    t0 = time.perf_counter()
    dat, ann = _cohort(layers, a.samples, dev, 1234) if rank == 0 else (None, None)
    dat, ann = trials.broadcast_cohort(dat, ann, dev)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    feats = {k: [f"{k}_{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(a.samples)], {})
    plist = trials.draw_search_space(a.trials, seed=a.seed, epochs=a.epochs)
    n_train = a.samples - int(a.samples * 0.2)
    costs = [trials.trial_cost(p, 2 * a.features, n_train) for p in plist]

    def trial_fn(tid, params):
        val, epochs, model, info = run_trial(DirectPred, params, ds, ["y"], early_stop_patience=0, seed=a.seed * 100003 + tid,
                                             device=dev)
        if "error" in info:
            raise RuntimeError(info["error"])
        trial_fn.samples += info["steps"] * int(params["batch_size"])
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        del model
        return val, epochs, sd
    trial_fn.samples = 0

    def shapes_of(params):
        return spec_from_dataset("DirectPred", params, ds, ["y"]).state_shapes()

    t1 = time.perf_counter()
    table, best, state = trials.run_sweep(plist, trial_fn, costs, dev, shapes_of)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t1
    tot = torch.tensor([float(trial_fn.samples), wall], dtype=torch.float64, device=dev)
    if world > 1:
        s = tot[:1].clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        w = tot[1:].clone()
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        tot = torch.cat([s, w])
    if rank == 0:
        ok = table[:, 3] == trials.STATUS_OK
        print(json.dumps({
            "workload": f"cfg5: {a.trials} DirectPred trials (2 x {a.features} features, N={a.samples}, {a.epochs} epochs), "
                        f"{world} GPU(s), trial sharding",
            "n_gpus": world, "trials": a.trials, "trials_ok": int(ok.sum()), "best_trial": best,
            "best_val_loss": float(table[best, 1]), "best_params": plist[best],
            "winner_state_tensors": len(state) if state is not None else 0,
            "cohort_broadcast_s": round(t_bcast, 4), "sweep_wall_s": round(float(tot[1]), 3),
            "aggregate_samples_per_s": round(float(tot[0]) / float(tot[1]), 1)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
