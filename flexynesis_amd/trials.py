"""Trial sharding across the GPUs of one node (BASELINE configs[4]).

The reference runs HPO trials strictly one after another on one device (reference main.py:352-368,
``devices=1`` at :223).  Trials share only the read-only cohort, so here they are the unit of
parallelism: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" for
the CPU tests), three collectives per sweep and NONE per training step:

  1. broadcast   rank 0 -> all: layer order, label names, then the cohort tensors (once)
  2. all_gather  fixed-size records (trial id, val_loss, epochs, status) after the local trials finish
  3. broadcast   winner rank -> all: the best trial's state_dict (flat fp32 buffer)

Gradient data-parallelism inside a trial is deliberately NOT used: all-reducing 0.8 GB of fp32 grads per
0.7-2.5 ms step over a per-link-bound xGMI ring would cost ~9 ms, and splitting the batch would change
BatchNorm batch statistics, i.e. break parity with the reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

STATUS_OK, STATUS_FAILED = 0.0, 1.0


def assign_trials(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of trials to ranks (trial cost ~ params/batch varies ~6x over
    the reference's search space, reference config.py:7-15).  Deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += float(costs[i])
    return [sorted(x) for x in out]


def trial_cost(params: dict, sum_features: int, n_train: int) -> float:
    """Predicted relative cost of one trial: HBM-bound step (~ parameter count) x optimisation steps."""
    p = sum_features * sum_features * float(params["hidden_dim_factor"])
    steps = max(n_train // int(params["batch_size"]), 1) * int(params.get("epochs", 1))
    return p * steps


def broadcast_cohort(dat: Optional[Dict[str, torch.Tensor]], ann: Optional[Dict[str, torch.Tensor]], device,
                     src: int = 0):
    """Rank ``src`` holds the cohort; every rank returns (dat, ann) as fp32 tensors on ``device`` with the key
    ORDER of rank ``src`` (modality order is hash-order dependent in the reference, data.py:508-515, so ranks
    must never recompute it)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ({k: v.to(device, torch.float32) for k, v in dat.items()},
                {k: v.to(device, torch.float32) for k, v in ann.items()})
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta = [{"dat": [(k, tuple(v.shape)) for k, v in dat.items()], "ann": [(k, tuple(v.shape)) for k, v in ann.items()]}]
    dist.broadcast_object_list(meta, src=src)
    out_d, out_a = {}, {}
    for group, src_dict, out in (("dat", dat, out_d), ("ann", ann, out_a)):
        for k, shp in meta[0][group]:
            if rank == src:
                t = src_dict[k].to(device, torch.float32).contiguous()
            else:
                t = torch.empty(shp, dtype=torch.float32, device=device)
            dist.broadcast(t, src=src)
            out[k] = t
    return out_d, out_a


def gather_results(local: List[Tuple[int, float, int, float]], n_trials: int, device) -> np.ndarray:
    """all_gather of (trial_id, val_loss, epochs, status) rows; returns an [n_trials, 4] array ordered by trial
    id.  A trial nobody reported is marked failed with val_loss=+inf (never a hang)."""
    table = torch.full((n_trials, 4), float("nan"), dtype=torch.float64, device=device)
    table[:, 0] = torch.arange(n_trials, device=device)
    table[:, 1] = float("inf")
    table[:, 3] = STATUS_FAILED
    mine = torch.full((n_trials, 4), float("nan"), dtype=torch.float64, device=device)
    for (tid, val, ep, status) in local:
        mine[tid] = torch.tensor([tid, val, ep, status], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    for p in parts:
        ok = ~torch.isnan(p[:, 0])
        table[ok] = p[ok]
    return table.cpu().numpy()


def broadcast_state(state: Optional[Dict[str, torch.Tensor]], shapes: Dict[str, tuple], src: int, device):
    """Winner's state_dict -> every rank, as one flat fp32 buffer (num_batches_tracked rides along as floats)."""
    keys = list(shapes.keys())
    sizes = [int(np.prod(shapes[k])) if shapes[k] else 1 for k in keys]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1 or dist.get_rank() == src:
        flat = torch.cat([state[k].detach().to(device, torch.float32).reshape(-1) for k in keys])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    out, o = {}, 0
    for k, n in zip(keys, sizes):
        t = flat[o:o + n].reshape(shapes[k])
        out[k] = t.to(torch.int64) if k.endswith("num_batches_tracked") else t
        o += n
    return out


def run_sweep(param_list: List[dict], trial_fn: Callable[[int, dict], Tuple[float, int, Optional[dict]]],
              costs: Optional[Sequence[float]] = None, device="cpu", state_shapes: Optional[Dict[str, tuple]] = None):
    """Shard ``param_list`` over the ranks, run ``trial_fn(trial_id, params) -> (val_loss, epochs, state_dict)``
    locally, gather the result table, and (if ``state_shapes`` is given) broadcast the winner's weights.
    ``state_shapes`` may be a dict (all trials share one architecture) or a callable ``params -> {key: shape}``
    (HPO: latent size / hidden factor differ per trial, so the winner's layout is derived from its parameters on
    every rank).  Returns (table [n,4], best_trial_id, best_state or None)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = len(param_list)
    costs = list(costs) if costs is not None else [1.0] * n
    assignment = assign_trials(costs, world)
    mine = assignment[rank]
    local = []
    best_local = (float("inf"), -1, None)          # only the best local trial's weights are kept (0.8 GB each at cfg2)
    for tid in mine:
        try:
            val, epochs, state = trial_fn(tid, param_list[tid])
            status = STATUS_OK if (val == val and math.isfinite(val)) else STATUS_FAILED
            val = val if status == STATUS_OK else float("inf")
        except Exception:                      # a broken trial reports +inf; the sweep goes on
            val, epochs, state, status = float("inf"), 0, None, STATUS_FAILED
        local.append((tid, float(val), int(epochs), status))
        # ties resolve to the lowest trial id, like np.argmin over the gathered table
        if state is not None and status == STATUS_OK and (best_local[2] is None or (val, tid) < best_local[:2]):
            best_local = (float(val), tid, state)
        del state
    table = gather_results(local, n, device)
    best = int(np.argmin(table[:, 1]))
    best_state = None
    if state_shapes is not None and math.isfinite(table[best, 1]):
        owner = next(r for r, lst in enumerate(assignment) if best in lst)
        # Every rank must take the same branch: the owner announces whether it actually holds the winner's weights
        # (a trial_fn may return None for them), so a missing state degrades to best_state=None everywhere instead of the
        # owner raising while the others wait in the broadcast.
        have = torch.tensor([1.0 if (rank == owner and best_local[1] == best and best_local[2] is not None) else 0.0],
                            dtype=torch.float32, device=device)
        if dist.is_initialized() and world > 1:
            dist.all_reduce(have, op=dist.ReduceOp.SUM)
        if float(have.item()) > 0:
            shapes = state_shapes(param_list[best]) if callable(state_shapes) else state_shapes
            best_state = broadcast_state(best_local[2] if rank == owner else None, shapes, owner, device)
    return table, best, best_state


def draw_search_space(n: int, seed: int = 0, epochs: int = 3) -> List[dict]:
    """n parameter dicts from the reference's DirectPred search space (reference config.py:7-15 + batch sizes
    of main.py:183-190), drawn with numpy's default_rng(seed) -- the cfg5 trial list of SURVEY.md section 8(d)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append({"latent_dim": int(rng.integers(16, 129)), "hidden_dim_factor": float(rng.uniform(0.2, 0.5)),
                    "lr": float(10 ** rng.uniform(-4, -2)), "supervisor_hidden_dim": int(rng.integers(8, 33)),
                    "batch_size": int(rng.choice([32, 64, 128])), "epochs": int(epochs)})
    return out
