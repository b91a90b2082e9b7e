"""ORACLE-side CPU baseline timer (test infrastructure; imported only by bench.py's cpu_baseline leg).

Times the reference's training loop as the reference runs it on CPU -- zero_grad -> training_step ->
backward -> clip_grad_norm_(1.0) -> torch.optim.Adam.step (reference main.py:212-225,
models/direct_pred.py:143,225-260) -- with the forward arithmetic taken from the pinned restatement
(oracle/restate.py) and torch's own in-place optimiser, i.e. the same ATen CPU kernels the reference
dispatches.  ``kind`` = "port": /root/reference does not exist on the GPU box.
"""
from __future__ import annotations

import time

import torch

from . import restate as O


def time_training(spec: O.Spec, n_samples: int, batch: int, steps: int, warmup: int = 1, threads: int = 0,
                  lr: float = 1e-3, seed: int = 1234):
    """Returns dict(samples_per_s, ms_per_step, steps, threads)."""
    if threads > 0:
        torch.set_num_threads(threads)
    torch.set_float32_matmul_precision("highest")
    dat, ann = O.synthetic_cohort(spec.layers, n_samples, seed)
    st = O.init_state(spec, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if not O.is_buffer(k)}
    work = dict(st)
    work.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    g = torch.Generator().manual_seed(seed + 1)
    times = []
    for s in range(warmup + steps):
        idx = torch.randperm(n_samples, generator=g)
        t0 = time.perf_counter()
        if spec.model == "MultiTripletNetwork":
            b = {"anchor": [dat[n][idx[:batch]] for n, _ in spec.layers],
                 "positive": [dat[n][idx[batch:2 * batch]] for n, _ in spec.layers],
                 "negative": [dat[n][idx[2 * batch:3 * batch]] for n, _ in spec.layers]}
        else:
            b = {"x": [dat[n][idx[:batch]] for n, _ in spec.layers]}
        b["y"] = {k: v[idx[:batch]] for k, v in ann.items()}
        draws = {}
        n = len(spec.layers)
        H = [spec.hidden(i) for i in range(n)]
        if spec.model == "supervised_vae":
            draws["eps"] = torch.randn(batch, spec.latent_dim)
            for i in range(n):
                draws[f"prior.{i}"] = torch.randn(O.MMD_PRIOR_SAMPLES, spec.latent_dim)
        else:
            for tag in (("@a", "@p", "@n") if spec.model == "MultiTripletNetwork" else ("",)):
                for i in range(n):
                    draws[f"encoders.{i}{tag}"] = torch.empty(batch, H[i]).bernoulli_(0.9)
        for (v, _, _) in spec.variables:
            draws["MLPs." + v] = torch.empty(batch, spec.sup_hidden).bernoulli_(0.9)
        opt.zero_grad(set_to_none=True)
        new_buffers = {}
        losses, _ = O.forward_losses(spec, work, b, True, draws, new_buffers)
        losses["total"].sum().backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        work.update(new_buffers)
        dt = time.perf_counter() - t0
        if s >= warmup:
            times.append(dt)
    tot = sum(times)
    return {"samples_per_s": batch * len(times) / tot, "ms_per_step": 1e3 * tot / len(times), "steps": len(times),
            "threads": torch.get_num_threads()}
