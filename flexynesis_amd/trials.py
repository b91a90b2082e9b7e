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

import contextlib
import math
import threading
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

STATUS_OK, STATUS_FAILED = 0.0, 1.0


def collective_device(device):
    """Where the tensors of a collective live: the GPU for RCCL ("nccl"); the host for gloo, which moves host memory (device tensors
    are staged through it around the call) -- the CPU tests, two ranks sharing one GPU in the gpu suite, FX_BENCH_BACKEND=gloo."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(device)


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


# Cost model of one engine fit on one MI355X, fitted to measured step times (profiles/r03_trial_breakdown.md):
# an optimisation step streams 24 B per wide-weight parameter at ~5 TB/s and has a latency-bound tail that does not
# depend on the batch size; a validation batch reads the wide weights once (4 B/param); every fit pays a fixed cost
# (model initialisation on the device, tape recording, two hipGraph captures).
COST_STEP_FIXED_S = 0.28e-3
COST_STEP_PER_PARAM_S = 24.0 / 5.0e12
COST_VAL_FIXED_S = 0.20e-3
COST_VAL_PER_PARAM_S = 4.0 / 3.5e12
COST_FIT_FIXED_S = 0.06


def trial_cost(params: dict, sum_features, n_train: int, n_val: int = 0) -> float:
    """Predicted seconds of one trial.  ``sum_features``: the layers' feature counts (a sequence), or their sum for
    equally wide layers (two, as in cfg2 / cfg5).  The step time barely depends on the batch size (1.24 -> 1.38 ms from
    B = 32 to 128 at cfg2), so the number of steps -- n_train // batch_size per epoch -- is what separates trials."""
    if isinstance(sum_features, (int, float)):
        feats = [float(sum_features) / 2.0] * 2
    else:
        feats = [float(f) for f in sum_features]
    p = sum(f * f * float(params["hidden_dim_factor"]) for f in feats)
    B = int(params["batch_size"])
    epochs = int(params.get("epochs", 1))
    steps = max(int(n_train) // B, 1)
    vals = -(-int(n_val) // B) if n_val else 0
    per_epoch = steps * (COST_STEP_FIXED_S + COST_STEP_PER_PARAM_S * p) + vals * (COST_VAL_FIXED_S + COST_VAL_PER_PARAM_S * p)
    return COST_FIT_FIXED_S + epochs * per_epoch + vals * (COST_VAL_FIXED_S + COST_VAL_PER_PARAM_S * p)


def broadcast_cohort(dat: Optional[Dict[str, torch.Tensor]], ann: Optional[Dict[str, torch.Tensor]], device,
                     src: int = 0, force_collectives: bool = False):
    """Rank ``src`` holds the cohort; every rank returns (dat, ann) as fp32 tensors on ``device`` with the key
    ORDER of rank ``src`` (modality order is hash-order dependent in the reference, data.py:508-515, so ranks
    must never recompute it)."""
    # force_collectives: take the collective branch on a one-rank group too (the world-1 GPU test then runs the very RCCL
    # calls an 8-GPU job makes, instead of the early-out)
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collectives):
        return ({k: v.to(device, torch.float32) for k, v in dat.items()},
                {k: v.to(device, torch.float32) for k, v in ann.items()})
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta = [{"dat": [(k, tuple(v.shape)) for k, v in dat.items()], "ann": [(k, tuple(v.shape)) for k, v in ann.items()]}]
    dist.broadcast_object_list(meta, src=src)
    out_d, out_a = {}, {}
    cdev = collective_device(device)
    for group, src_dict, out in (("dat", dat, out_d), ("ann", ann, out_a)):
        for k, shp in meta[0][group]:
            if rank == src:
                t = src_dict[k].to(cdev, torch.float32).contiguous()
            else:
                t = torch.empty(shp, dtype=torch.float32, device=cdev)
            dist.broadcast(t, src=src)
            out[k] = t.to(device)
    return out_d, out_a


def gather_results(local: List[Tuple[int, float, int, float]], n_trials: int, device, rank: int = 0,
                   force_collectives: bool = False) -> np.ndarray:
    """all_gather of (trial_id, val_loss, epochs, status, rank) rows; returns an [n_trials, 5] array ordered by trial
    id.  A trial nobody reported is marked failed with val_loss=+inf (never a hang)."""
    device = collective_device(device)
    table = torch.full((n_trials, 5), float("nan"), dtype=torch.float64, device=device)
    table[:, 0] = torch.arange(n_trials, device=device)
    table[:, 1] = float("inf")
    table[:, 2] = 0.0                 # a unit nobody reported: failed, +inf, 0 epochs, no owner (consumers take means of this column)
    table[:, 3] = STATUS_FAILED
    table[:, 4] = -1.0
    mine = torch.full((n_trials, 5), float("nan"), dtype=torch.float64, device=device)
    for (tid, val, ep, status) in local:
        mine[tid] = torch.tensor([tid, val, ep, status, rank], dtype=torch.float64, device=device)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if dist.is_initialized() and (world > 1 or force_collectives):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    for p in parts:
        ok = ~torch.isnan(p[:, 0])
        table[ok] = p[ok]
    return table.cpu().numpy()


def broadcast_state(state: Optional[Dict[str, torch.Tensor]], shapes: Dict[str, tuple], src: int, device,
                    force_collectives: bool = False):
    """Winner's state_dict -> every rank, as one flat fp32 buffer (num_batches_tracked rides along as floats)."""
    keys = list(shapes.keys())
    sizes = [int(np.prod(shapes[k])) if shapes[k] else 1 for k in keys]
    cdev = collective_device(device)
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=cdev)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if not dist.is_initialized() or world == 1 or dist.get_rank() == src:
        flat = torch.cat([state[k].detach().to(cdev, torch.float32).reshape(-1) for k in keys])
    if dist.is_initialized() and (world > 1 or force_collectives):
        dist.broadcast(flat, src=src)
    flat = flat.to(device)
    out, o = {}, 0
    for k, n in zip(keys, sizes):
        t = flat[o:o + n].reshape(shapes[k])
        out[k] = t.to(torch.int64) if k.endswith("num_batches_tracked") else t
        o += n
    return out


def _default_store():
    """The process group's rendezvous store (TCP / file store on the host): ``store.add`` is an atomic fetch-and-add
    across ranks, independent of the collective backend -- the shared counter of the work queue."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        return _get_default_store()
    except Exception:
        return None


def _agree_on_queue(store, order, costs) -> str:
    """The name of this run_units call's shared counter: rank 0 draws a fresh id from the store and BROADCASTS it, together with
    the number of units and a digest of the claim order, and every rank checks that it was about to run the same sweep.  (Round 3
    named the queue by a per-process call counter: a rank that had called run_units once more or once less -- an exception path,
    a rank-0-only warm-up -- silently split the queue.  Now such a rank fails the check, on every rank, before any unit runs.)"""
    import hashlib
    digest = hashlib.sha1(repr((len(order), list(order), [round(float(c), 9) for c in costs])).encode()).hexdigest()
    msg = [None]
    if dist.get_rank() == 0:
        msg = [{"sid": int(store.add("fx_amd/sweep_seq", 1)), "digest": digest}]
    dist.broadcast_object_list(msg, src=0)
    same = torch.tensor([1.0 if msg[0]["digest"] == digest else 0.0],
                        device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if float(same.item()) < 1.0:
        raise RuntimeError("run_units: the ranks do not agree on the units of this sweep (number, costs or order differ): "
                           "every rank must call run_units with the same arguments, the same number of times")
    return f"fx_amd/queue/{msg[0]['sid']}"


class _Claims:
    """Hands out unit indices to this rank: ``static`` = the deterministic LPT assignment computed on every rank,
    ``queue`` = units in longest-first order claimed one by one from a counter shared by all ranks (no rank idles while
    another still holds queued units: the tail imbalance is at most one unit).  Both modes visit every unit exactly
    once across the ranks; which rank ran a unit is reported in the gathered table, never assumed."""

    def __init__(self, costs: Sequence[float], world: int, rank: int, mode: str):
        self.order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
        self.mode, self.store = mode, None
        if mode == "queue" and world > 1:
            self.store = _default_store()
        if self.store is None:
            self.mode = "static"
            self.mine = list(assign_trials(costs, world)[rank])
        else:
            self.key = _agree_on_queue(self.store, self.order, costs)

    def __iter__(self):
        if self.mode == "static":
            yield from self.mine
            return
        while True:
            k = int(self.store.add(self.key, 1)) - 1
            if k >= len(self.order):
                return
            yield self.order[k]


LAST_ERRORS: Dict[int, str] = {}      # unit id -> repr of the exception that made it report +inf (this rank, the most recent run_units call)


def run_units(n: int, unit_fn: Callable[[int], Tuple[float, int, Optional[dict]]], costs: Optional[Sequence[float]] = None,
              device="cpu", keep: Optional[Sequence[int]] = None, schedule: str = "queue", force_collectives: bool = False,
              in_flight: int = 1, unit_bytes: Optional[float] = None, timings: Optional[dict] = None):
    """Run units 0..n-1 (HPO trials, cross-validation folds, fine-tuning fits) sharded over the ranks:
    ``unit_fn(uid) -> (val_loss, epochs, state_dict | None)``.  Returns (table [n, 5]: uid, val_loss, epochs, status,
    rank that ran it; local: {uid: state} of the units this rank must hold on to).  ``keep`` = unit ids whose state is
    needed afterwards whatever their loss (the FineTuner continues from its LAST fit, main.py:647); None keeps only this
    rank's best unit (0.8 GB of weights per cfg2 trial).  A failing / non-finite unit reports +inf and the sweep goes on.

    ``in_flight`` > 1: that many units run CONCURRENTLY on this rank's GPU, each on a host thread with its own HIP stream
    (eager launches: the units must not capture hipGraphs -- ``fit(use_graph=False)``).  One unit's latency-bound launches
    then run while the other's HBM-bound dW + Adam launches hold the memory system: +10 % aggregate samples/s at two units
    in flight on cfg5-style trials (scripts/bench_two_trials.py); the units' results are unchanged (every unit is seeded
    on its own and deterministic).  ``unit_bytes``: device memory the LARGEST unit needs (weights, Adam moments, activations);
    the number of units in flight is lowered until that many of them fit into 80 % of what is free now.
    ``timings``, if given, receives {"units_s", "gather_s"} of this rank.  The exception behind a unit that reported +inf is kept in
    ``trials.LAST_ERRORS`` (and warned about once per call)."""
    import time as _time
    import warnings
    LAST_ERRORS.clear()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    costs = list(costs) if costs is not None else [1.0] * n
    if len(costs) != n:
        raise ValueError(f"run_units: {len(costs)} costs for {n} units")
    if schedule not in ("queue", "static"):
        raise ValueError("schedule must be 'queue' or 'static'")
    keep_set = None if keep is None else set(int(k) for k in keep)
    local, held = [], {}
    best_local = [float("inf"), -1]
    claims = iter(_Claims(costs, world, rank, schedule))
    lock = threading.Lock()
    dev = torch.device(device)
    if int(in_flight) > 1 and unit_bytes and dev.type == "cuda":
        free_b = torch.cuda.mem_get_info(dev)[0]
        try:      # memory this process already holds for wide weights (engine.PartitionArena's pools) is invisible to the driver's figure
            from .engine import placement_memory
            free_b += placement_memory(dev)["arena_free_bytes"]
        except Exception:
            pass
        fit_n = max(int(0.8 * free_b // max(float(unit_bytes), 1.0)), 1)
        if fit_n < int(in_flight):
            warnings.warn(f"run_units: {in_flight} units in flight need ~{in_flight * unit_bytes / 2**30:.1f} GiB, {free_b / 2**30:.1f} GiB "
                          f"are free: running {fit_n} at a time", RuntimeWarning, stacklevel=2)
            in_flight = fit_n
    t_units = _time.perf_counter()

    def worker(own_stream: bool):
        if dev.type == "cuda":
            torch.cuda.set_device(dev)                      # (the current device is per host thread)
        if own_stream and dev.type == "cuda":
            from . import ops as _ops
            own = torch.cuda.Stream(dev)
            ctx = torch.cuda.stream(own)
            scope = _ops.in_flight_scope(own)      # no hipGraph capture on this thread; its side streams are released at the end
        else:
            ctx = scope = contextlib.nullcontext()
        with scope, ctx:
            while True:
                with lock:
                    uid = next(claims, None)
                if uid is None:
                    return
                try:
                    val, epochs, state = unit_fn(uid)
                    status = STATUS_OK if (val == val and math.isfinite(val)) else STATUS_FAILED
                    val = val if status == STATUS_OK else float("inf")
                except Exception as e:                 # a broken unit reports +inf; the sweep goes on
                    val, epochs, state, status = float("inf"), 0, None, STATUS_FAILED
                    with lock:
                        LAST_ERRORS[uid] = repr(e)
                with lock:
                    local.append((uid, float(val), int(epochs), status))
                    if state is not None and status == STATUS_OK:
                        if keep_set is not None:
                            if uid in keep_set:
                                held[uid] = state
                        elif best_local[1] < 0 or (val, uid) < tuple(best_local):   # ties resolve to the lowest id, like np.argmin
                            held.clear()
                            held[uid] = state
                            best_local[0], best_local[1] = float(val), uid
                del state

    if int(in_flight) <= 1:
        worker(False)
    else:
        threads = [threading.Thread(target=worker, args=(True,), name=f"fx-unit-{i}") for i in range(int(in_flight))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if LAST_ERRORS:
        uid0 = sorted(LAST_ERRORS)[0]
        warnings.warn(f"run_units: {len(LAST_ERRORS)} unit(s) failed on rank {rank} and report +inf; unit {uid0}: {LAST_ERRORS[uid0]}",
                      RuntimeWarning, stacklevel=2)
    t_gather = _time.perf_counter()
    table = gather_results(local, n, device, rank, force_collectives)
    if timings is not None:
        timings["units_s"] = round(t_gather - t_units, 4)
        timings["gather_s"] = round(_time.perf_counter() - t_gather, 4)
    return table, held


def agree_and_broadcast_state(held: Dict[int, dict], uid: int, table: np.ndarray, shapes: Dict[str, tuple], device,
                              force_collectives: bool = False):
    """Unit ``uid``'s state_dict on every rank, or None everywhere when its owner does not hold it.  Every rank must take
    the same branch: the owner first announces whether it actually holds the weights (a unit_fn may return None), so a
    missing state degrades to None on all ranks instead of the owner raising while the others wait in the broadcast."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    owner = int(table[uid, 4]) if (table[uid, 4] == table[uid, 4] and table[uid, 4] >= 0) else 0
    have = torch.tensor([1.0 if (rank == owner and uid in held) else 0.0], dtype=torch.float32, device=collective_device(device))
    if dist.is_initialized() and (world > 1 or force_collectives):
        dist.all_reduce(have, op=dist.ReduceOp.SUM)
    if float(have.item()) <= 0:
        return None
    return broadcast_state(held.get(uid) if rank == owner else None, shapes, owner, device, force_collectives)


def run_sweep(param_list: List[dict], trial_fn: Callable[[int, dict], Tuple[float, int, Optional[dict]]],
              costs: Optional[Sequence[float]] = None, device="cpu", state_shapes: Optional[Dict[str, tuple]] = None,
              schedule: str = "queue", force_collectives: bool = False, in_flight: int = 1, unit_bytes: Optional[float] = None,
              timings: Optional[dict] = None):
    """Shard ``param_list`` over the ranks, run ``trial_fn(trial_id, params) -> (val_loss, epochs, state_dict)``
    locally, gather the result table, and (if ``state_shapes`` is given) broadcast the winner's weights.
    ``state_shapes`` may be a dict (all trials share one architecture) or a callable ``params -> {key: shape}``
    (HPO: latent size / hidden factor differ per trial, so the winner's layout is derived from its parameters on
    every rank).  Returns (table [n, 5], best_trial_id, best_state or None)."""
    n = len(param_list)
    import time as _time
    table, held = run_units(n, lambda uid: trial_fn(uid, param_list[uid]), costs, device, keep=None, schedule=schedule,
                            force_collectives=force_collectives, in_flight=in_flight, unit_bytes=unit_bytes, timings=timings)
    best = int(np.argmin(table[:, 1]))
    best_state = None
    t0 = _time.perf_counter()
    if state_shapes is not None and math.isfinite(table[best, 1]):
        shapes = state_shapes(param_list[best]) if callable(state_shapes) else state_shapes
        best_state = agree_and_broadcast_state(held, best, table, shapes, device, force_collectives)
    if timings is not None:
        timings["winner_broadcast_s"] = round(_time.perf_counter() - t0, 4)
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
