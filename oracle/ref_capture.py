"""Run the REAL reference model (build container only) and capture everything a golden needs.

Test infrastructure (see oracle/restate.py header for who may import ``oracle/``).  The reference's
own code is executed unmodified; the only interposition is *recording* wrappers around
``torch.nn.functional.dropout``, ``torch.randn_like`` and ``torch.randn`` that call the original
function and remember what it returned, so that the random draws of a step become explicit fixture
inputs (dropout masks, reparameterisation eps, the 200 MMD prior samples per modality).
"""
from __future__ import annotations

import contextlib
from types import SimpleNamespace
from typing import Dict, List

import torch

from . import ref_shim
from .restate import Spec


class Capture:
    def __init__(self):
        self.dropout_masks: List[torch.Tensor] = []
        self.randn_like: List[torch.Tensor] = []
        self.randn: List[torch.Tensor] = []


@contextlib.contextmanager
def capture_rng():
    import torch.nn.functional as F
    cap = Capture()
    o_drop, o_rl, o_rn = F.dropout, torch.randn_like, torch.randn

    def drop(input, p=0.5, training=True, inplace=False):
        out = o_drop(input, p, training, inplace)
        if training and p > 0:
            # out = input * (mask/(1-p)); where input == 0 the mask is unobservable AND irrelevant
            mask = torch.where(input != 0, (out != 0), torch.ones_like(out, dtype=torch.bool))
            cap.dropout_masks.append(mask.to(torch.float32).detach().clone())
        return out

    def rl(*a, **k):
        t = o_rl(*a, **k)
        cap.randn_like.append(t.detach().clone())
        return t

    def rn(*a, **k):
        t = o_rn(*a, **k)
        cap.randn.append(t.detach().clone())
        return t

    F.dropout, torch.randn_like, torch.randn = drop, rl, rn
    try:
        yield cap
    finally:
        F.dropout, torch.randn_like, torch.randn = o_drop, o_rl, o_rn


def make_dataset(R, dat: Dict[str, torch.Tensor], ann: Dict[str, torch.Tensor], variable_types):
    n = next(iter(dat.values())).shape[0]
    features = {k: [f"{k}_{j}" for j in range(v.shape[1])] for k, v in dat.items()}
    samples = [f"s{i}" for i in range(n)]
    return R.MultiOmicDataset(dat, ann, variable_types, features, samples, {})


def build_reference_model(R, spec: Spec, dataset, config: dict):
    cls = {"DirectPred": R.DirectPred, "supervised_vae": R.supervised_vae,
           "MultiTripletNetwork": R.MultiTripletNetwork, "CrossModalPred": R.CrossModalPred}[spec.model]
    targets = [v[0] for v in spec.variables if v[0] != spec.surv_event_var]
    extra = {}
    if spec.model == "CrossModalPred":          # crossmodal_pred.py:39-40
        extra = dict(input_layers=spec.input_layers, output_layers=spec.output_layers)
    model = cls(config, dataset, targets, batch_variables=None,
                surv_event_var=spec.surv_event_var, surv_time_var=spec.surv_time_var,
                use_loss_weighting=spec.use_loss_weighting, device_type="cpu", **extra)
    return model


def name_draws(spec: Spec, cap: Capture) -> Dict[str, torch.Tensor]:
    """Map the capture order to names (RNG consumption order)."""
    draws: Dict[str, torch.Tensor] = {}
    masks = list(cap.dropout_masks)
    n = len(spec.layers)
    if spec.model == "DirectPred":
        for i in range(n):
            draws[f"encoders.{i}"] = masks.pop(0)
    elif spec.model == "MultiTripletNetwork":
        for tag in ("@a", "@p", "@n"):
            for i in range(n):
                draws[f"encoders.{i}{tag}"] = masks.pop(0)
    else:
        draws["eps"] = cap.randn_like[0]
        for j in range(len(spec.dec_idx)):      # one 200-sample prior draw per reconstructed layer
            draws[f"prior.{j}"] = cap.randn[j]
    for (v, _, _) in spec.variables:
        draws["MLPs." + v] = masks.pop(0)
    assert not masks, "unconsumed dropout draws"
    return draws


def reference_batch(spec: Spec, batch):
    first = batch["anchor"][0] if spec.model == "MultiTripletNetwork" else batch["x"][0]
    samples = tuple(f"s{i}" for i in range(first.shape[0]))          # (no label dict entry exists in unsupervised runs)
    if spec.model == "MultiTripletNetwork":
        d = lambda xs: {name: x for (name, _), x in zip(spec.layers, xs)}
        return (d(batch["anchor"]), d(batch["positive"]), d(batch["negative"]), batch["y"])
    return ({name: x for (name, _), x in zip(spec.layers, batch["x"])}, batch["y"], samples)


def reference_train_steps(R, spec: Spec, model, batches, lr: float, clip: bool = True, freeze=None):
    """zero_grad -> training_step -> backward -> clip_grad_norm_(1.0) -> Adam.step per batch,
    exactly what Lightning's automatic optimisation does with the Trainer of reference main.py:212-225.
    ``freeze`` ({"encoders": bool, "supervisors": bool}) + ``clip=False`` reproduce the FineTuner's setup instead
    (main.py:530-539 requires_grad flags, :562-566 Adam over the trainable parameters, :591-600 a Trainer without
    gradient clipping).  Returns a list of per-step records."""
    if freeze is not None:
        for enc in model.encoders:
            for p in enc.parameters():
                p.requires_grad = not freeze["encoders"]
        for mlp in model.MLPs.values():
            for p in mlp.parameters():
                p.requires_grad = not freeze["supervisors"]
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr)
    model.train()
    recs = []
    for bi, batch in enumerate(batches):
        opt.zero_grad()
        with capture_rng() as cap:
            loss = model.training_step(reference_batch(spec, batch), bi, log=False)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        if clip:
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        else:
            gn = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
        opt.step()
        recs.append(SimpleNamespace(
            total=loss.detach().clone(), grads=grads, grad_norm=gn.detach().clone(),
            draws=name_draws(spec, cap),
            state={k: v.detach().clone() for k, v in model.state_dict().items()},
            exp_avg={k: opt.state[p]["exp_avg"].detach().clone() for k, p in model.named_parameters() if p in opt.state},
            exp_avg_sq={k: opt.state[p]["exp_avg_sq"].detach().clone() for k, p in model.named_parameters() if p in opt.state},
        ))
    return recs
