"""Generate the golden vectors in tests/golden/ FROM THE REFERENCE ITSELF (build container only).

    python -m oracle.gen_goldens            # rewrites tests/golden/*.npz

Each golden holds: the architecture spec, an initial state_dict, K batches, the random draws the
reference consumed in each step (recorded, see oracle/ref_capture.py), and what the reference
produced: every named loss, every gradient, the global grad norm, and the full state_dict plus Adam
moments after each optimiser step.  Function-level goldens (cox / mse / ce / mmd / triplet edge
cases, eval-mode predict/transform) are in ``functions.npz``.

A golden is DATA (inputs and expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import dataclasses
import json
import os
import sys

import numpy as np
import torch

from . import ref_capture, ref_shim
from .restate import Spec, init_state, state_manifest

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def case_specs():
    """name -> (spec, config, n_samples, batch_size, n_steps, label tweaks)"""
    c = {}
    c["directpred_2omics_multitask"] = dict(
        spec=Spec("DirectPred", [("gex", 64), ("cnv", 48)], 8, 0.25, 4,
                  [("y", "numerical", 1), ("c", "categorical", 3), ("event", "numerical", 1)],
                  surv_event_var="event", surv_time_var="time"),
        n=40, B=8, steps=3, missing=True)
    c["directpred_1omics_regression"] = dict(       # BASELINE cfg1 shape family: no fusion, 1 loss term
        spec=Spec("DirectPred", [("gex", 80)], 8, 0.25, 4, [("y", "numerical", 1)]),
        n=32, B=8, steps=3, missing=False)
    c["directpred_unweighted"] = dict(
        spec=Spec("DirectPred", [("gex", 40), ("cnv", 24)], 6, 0.3, 5,
                  [("y", "numerical", 1), ("c", "categorical", 4)], use_loss_weighting=False),
        n=24, B=12, steps=2, missing=True)
    c["supervised_vae_2omics"] = dict(
        spec=Spec("supervised_vae", [("gex", 56), ("cnv", 40)], 8, 0.25, 4,
                  [("c", "categorical", 3), ("event", "numerical", 1)],
                  surv_event_var="event", surv_time_var="time"),
        n=40, B=10, steps=3, missing=True)
    c["crossmodal_2in_2out"] = dict(               # encode (gex, cnv), reconstruct (meth, gex)
        spec=Spec("CrossModalPred", [("gex", 48), ("cnv", 40), ("meth", 36)], 8, 0.25, 4,
                  [("y", "numerical", 1), ("c", "categorical", 3)],
                  input_layers=["gex", "cnv"], output_layers=["meth", "gex"]),
        n=40, B=10, steps=3, missing=True)
    c["triplet_3omics"] = dict(
        spec=Spec("MultiTripletNetwork", [("gex", 48), ("cnv", 40), ("meth", 32)], 8, 0.25, 4,
                  [("c", "categorical", 3), ("y", "numerical", 1)]),
        n=36, B=8, steps=2, missing=False)
    return c


def make_cohort(spec: Spec, n: int, seed: int, missing: bool):
    g = torch.Generator().manual_seed(seed)
    dat = {name: torch.randn(n, F, generator=g) for name, F in spec.layers}
    ann, vt = {}, {}
    for (v, kind, C) in spec.variables:
        if v == spec.surv_event_var:
            ann[v] = (torch.rand(n, generator=g) < 0.6).float()
            ann[spec.surv_time_var] = torch.rand(n, generator=g) * 10     # tie-free
            vt[v] = "numerical"
            vt[spec.surv_time_var] = "numerical"
        elif kind == "numerical":
            ann[v] = torch.randn(n, generator=g)
            vt[v] = "numerical"
        else:
            lab = torch.arange(n) % C                 # every class present -> C = len(unique)
            ann[v] = lab[torch.randperm(n, generator=g)].float()
            vt[v] = "categorical"
    return dat, ann, vt


def make_batches(spec: Spec, dat, ann, B, steps, seed, missing):
    g = torch.Generator().manual_seed(seed + 1)
    n = next(iter(dat.values())).shape[0]
    batches = []
    for s in range(steps):
        idx = torch.randperm(n, generator=g)[:B]
        y = {k: v[idx].clone() for k, v in ann.items()}
        if missing:
            for (v, kind, C) in spec.variables:
                if v == spec.surv_event_var:
                    y[v][0] = float("nan")
                    y[spec.surv_time_var][1] = float("nan")
                elif kind == "numerical":
                    y[v][2] = float("nan")
                else:
                    y[v][3] = float("nan")
                    y[v][4] = -1.0
        if spec.model == "MultiTripletNetwork":
            ip = torch.randperm(n, generator=g)[:B]
            ineg = torch.randperm(n, generator=g)[:B]
            batches.append({"anchor": [dat[k][idx] for k, _ in spec.layers],
                            "positive": [dat[k][ip] for k, _ in spec.layers],
                            "negative": [dat[k][ineg] for k, _ in spec.layers], "y": y})
        else:
            batches.append({"x": [dat[k][idx] for k, _ in spec.layers], "y": y})
    return batches


def perturbed_state(spec: Spec, seed: int):
    """init_state + non-trivial BN affine / running stats / log_vars so nothing hides behind 0/1."""
    st = init_state(spec, seed)
    g = torch.Generator().manual_seed(seed + 7)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = torch.randn(st[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        elif (".batchnorm." in k or ".hidden_layers.2." in k) and k.endswith("weight"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        elif (".batchnorm." in k or ".hidden_layers.2." in k) and k.endswith("bias"):
            st[k] = torch.randn(st[k].shape, generator=g) * 0.1
        elif k.startswith("log_vars."):
            st[k] = torch.randn(st[k].shape, generator=g) * 0.2
    return st


def gen_model_case(R, name, cfg, out):
    spec: Spec = cfg["spec"]
    lr = 1e-2
    dat, ann, vt = make_cohort(spec, cfg["n"], seed=100, missing=cfg["missing"])
    ds = ref_capture.make_dataset(R, dat, ann, vt)
    config = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": lr,
              "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": 1, "batch_size": cfg["B"]}
    torch.manual_seed(11)
    model = ref_capture.build_reference_model(R, spec, ds, config)
    ref_keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ref_keys == state_manifest(spec), (name, set(ref_keys) ^ set(state_manifest(spec)))
    st0 = perturbed_state(spec, seed=5)
    model.load_state_dict(st0)
    batches = make_batches(spec, dat, ann, cfg["B"], cfg["steps"], seed=200, missing=cfg["missing"])
    logged = []
    model.log_dict = lambda d, **k: logged.append({a: b.detach().clone() for a, b in d.items()})
    # reference_train_steps calls training_step(log=False); re-enable logging to capture named losses
    orig_ts = model.training_step
    model.training_step = lambda b, i, log=True: orig_ts(b, i, log=True)
    torch.manual_seed(12)
    recs = ref_capture.reference_train_steps(R, spec, model, batches, lr)
    arrays = {"spec_json": np.array(json.dumps(dataclasses.asdict(spec))), "lr": np.array(lr),
              "n_steps": np.array(len(batches))}
    for k, v in st0.items():
        arrays["state0/" + k] = _np(v)
    for s, (b, r, lg) in enumerate(zip(batches, recs, logged)):
        for part in ("x", "anchor", "positive", "negative"):
            if part in b:
                for i, x in enumerate(b[part]):
                    arrays[f"batch/{s}/{part}/{i}"] = _np(x)
        for k, v in b["y"].items():
            arrays[f"batch/{s}/y/{k}"] = _np(v)
        for k, v in r.draws.items():
            arrays[f"draws/{s}/{k}"] = _np(v)
        for k, v in lg.items():
            arrays[f"exp/{s}/loss/{'total' if k == 'train_loss' else k}"] = _np(v).reshape(-1)[0:1].reshape(())
        for k, v in r.grads.items():
            arrays[f"exp/{s}/grad/{k}"] = _np(v)
        arrays[f"exp/{s}/grad_norm"] = _np(r.grad_norm)
        for k, v in r.state.items():
            arrays[f"exp/{s}/state/{k}"] = _np(v)
        for k, v in r.exp_avg.items():
            arrays[f"exp/{s}/m/{k}"] = _np(v)
        for k, v in r.exp_avg_sq.items():
            arrays[f"exp/{s}/v/{k}"] = _np(v)
    # eval-mode outputs after training: validation_step total + predict/transform arithmetic
    model.eval()
    vb = batches[0]
    with torch.no_grad(), ref_capture.capture_rng() as cap:
        logged.clear()
        model.validation_step(ref_capture.reference_batch(spec, vb), 0, log=True)
    for k, v in logged[0].items():
        arrays[f"exp/val/loss/{'total' if k == 'val_loss' else k}"] = _np(v).reshape(-1)[0:1].reshape(())
    if spec.is_vae:
        for j in range(len(spec.dec_idx)):
            arrays[f"draws/val/prior.{j}"] = _np(cap.randn[j])
        arrays["draws/val/eps"] = _np(cap.randn_like[0])
    if spec.model != "MultiTripletNetwork":   # triplet transform/predict need the full dataset object
        with torch.no_grad(), ref_capture.capture_rng() as cap_p:
            pred = model.predict(ds)
        with torch.no_grad(), ref_capture.capture_rng() as cap_t:
            emb = model.transform(ds)
        if spec.is_vae:
            # the reference's predict/transform run the full stochastic forward even in eval mode
            # (supervised_vae.py:417-419, :470): z = mean + log_var * randn_like(log_var)
            arrays["draws/predict/eps"] = _np(cap_p.randn_like[0])
            arrays["draws/transform/eps"] = _np(cap_t.randn_like[0])
        for k, v in pred.items():
            arrays[f"exp/predict/{k}"] = np.asarray(v, dtype=np.float32)
        arrays["exp/transform"] = emb.values.astype(np.float32)
        for k, v in dat.items():
            arrays["cohort/" + k] = _np(v)
    np.savez_compressed(os.path.join(out, name + ".npz"), **arrays)
    print(f"[golden] {name}: {len(arrays)} arrays, total loss per step =",
          [float(r.total.reshape(-1)[0]) for r in recs])


def gen_function_goldens(R, out):
    """Edge cases of the loss functions, straight from the reference's own functions."""
    g = torch.Generator().manual_seed(77)
    A = {}

    def cox_case(tag, o, d, e):
        o = o.clone().requires_grad_(True)
        l = R.cox_ph_loss(o, d, e)
        A[f"cox/{tag}/outputs"], A[f"cox/{tag}/durations"], A[f"cox/{tag}/events"] = _np(o), _np(d), _np(e)
        A[f"cox/{tag}/loss"] = _np(l)
        if l.requires_grad and l.grad_fn is not None:
            l.backward()
            A[f"cox/{tag}/grad"] = _np(o.grad)
        else:
            A[f"cox/{tag}/grad"] = np.zeros(tuple(o.shape), np.float32)

    n = 12
    o = torch.randn(n, 1, generator=g)
    d = torch.rand(n, generator=g) * 5
    e = (torch.rand(n, generator=g) < 0.5).float()
    e[0] = 1.0
    cox_case("plain", o, d, e)
    d2, e2 = d.clone(), e.clone()
    d2[3] = float("nan")
    e2[5] = float("nan")
    cox_case("with_nan", o, d2, e2)
    cox_case("all_censored", o, d, torch.zeros(n))                      # 0/0 -> non-finite -> 0
    d3 = torch.full((n,), float("nan"))
    cox_case("none_valid", o, d3, e)
    d4, e4 = torch.full((n,), float("nan")), torch.full((n,), float("nan"))
    d4[7], e4[7] = 2.0, 1.0
    cox_case("single_valid", o, d4, e4)
    cox_case("large_batch", torch.randn(128, 1, generator=g), torch.rand(128, generator=g) * 9,
             (torch.rand(128, generator=g) < 0.5).float())

    # compute_loss (MSE / CE) incl. all-missing; borrow a DirectPred instance for the bound method
    spec = Spec("DirectPred", [("gex", 16)], 4, 0.5, 4, [("y", "numerical", 1), ("c", "categorical", 3)])
    dat, ann, vt = make_cohort(spec, 12, 1, False)
    ds = ref_capture.make_dataset(R, dat, ann, vt)
    m = ref_capture.build_reference_model(R, spec, ds, {"latent_dim": 4, "hidden_dim_factor": 0.5, "lr": 1e-3,
                                                         "supervisor_hidden_dim": 4, "epochs": 1, "batch_size": 4})

    def loss_case(tag, var, y, yhat):
        yhat = yhat.clone().requires_grad_(True)
        l = m.compute_loss(var, y, yhat)
        A[f"{tag}/y"], A[f"{tag}/yhat"], A[f"{tag}/loss"] = _np(y), _np(yhat), _np(l)
        if l.grad_fn is not None:
            l.backward()
            A[f"{tag}/grad"] = _np(yhat.grad)
        else:
            A[f"{tag}/grad"] = np.zeros(tuple(yhat.shape), np.float32)

    y = torch.randn(10, generator=g)
    yh = torch.randn(10, 1, generator=g)
    loss_case("mse/plain", "y", y, yh)
    y2 = y.clone()
    y2[[1, 4]] = float("nan")
    loss_case("mse/with_nan", "y", y2, yh)
    loss_case("mse/all_missing", "y", torch.full((10,), float("nan")), yh)
    c = torch.randint(0, 3, (10,), generator=g).float()
    lg = torch.randn(10, 3, generator=g) * 2
    loss_case("ce/plain", "c", c, lg)
    c2 = c.clone()
    c2[0] = -1.0
    c2[6] = float("nan")
    loss_case("ce/with_missing", "c", c2, lg)
    loss_case("ce/all_missing", "c", torch.full((10,), -1.0), lg)

    # total-loss combos
    m.log_vars["y"].data.fill_(0.3)
    m.log_vars["c"].data.fill_(-0.2)
    l1, l2 = torch.tensor(1.7), torch.tensor(0.4)
    A["total/weighted"] = _np(m.compute_total_loss({"y": l1, "c": l2}))
    A["total/single"] = _np(m.compute_total_loss({"y": l1}))
    A["total/l1"], A["total/l2"] = _np(l1), _np(l2)
    A["total/s_y"], A["total/s_c"] = np.float32(0.3), np.float32(-0.2)

    # triplet + MMD
    spec3 = Spec("MultiTripletNetwork", [("gex", 16)], 6, 0.5, 4, [("c", "categorical", 3)])
    dat3, ann3, vt3 = make_cohort(spec3, 12, 2, False)
    ds3 = ref_capture.make_dataset(R, dat3, ann3, vt3)
    t = ref_capture.build_reference_model(R, spec3, ds3, {"latent_dim": 6, "hidden_dim_factor": 0.5, "lr": 1e-3,
                                                           "supervisor_hidden_dim": 4, "epochs": 1, "batch_size": 4})
    a, p, ng = (torch.randn(9, 6, generator=g).requires_grad_(True) for _ in range(3))
    l = t.triplet_loss(a, p, ng)
    l.backward()
    A["triplet/a"], A["triplet/p"], A["triplet/n"], A["triplet/loss"] = _np(a), _np(p), _np(ng), _np(l)
    A["triplet/grad_a"], A["triplet/grad_p"], A["triplet/grad_n"] = _np(a.grad), _np(p.grad), _np(ng.grad)

    specv = Spec("supervised_vae", [("gex", 16)], 5, 0.5, 4, [("c", "categorical", 3)])
    datv, annv, vtv = make_cohort(specv, 12, 3, False)
    dsv = ref_capture.make_dataset(R, datv, annv, vtv)
    v = ref_capture.build_reference_model(R, specv, dsv, {"latent_dim": 5, "hidden_dim_factor": 0.5, "lr": 1e-3,
                                                           "supervisor_hidden_dim": 4, "epochs": 1, "batch_size": 4})
    z = torch.randn(7, 5, generator=g).requires_grad_(True)
    x = torch.randn(7, 16, generator=g)
    xh = torch.rand(7, 16, generator=g).requires_grad_(True)
    with ref_capture.capture_rng() as cap:
        l = v.MMD_loss(5, z, xh, x)
    l.backward()
    A["mmd/z"], A["mmd/x"], A["mmd/xhat"], A["mmd/prior"] = _np(z), _np(x), _np(xh), _np(cap.randn[0])
    A["mmd/loss"], A["mmd/grad_z"], A["mmd/grad_xhat"] = _np(l), _np(z.grad), _np(xh.grad)
    A["mmd/kernel_zz"] = _np(v.compute_kernel(z.detach(), z.detach()))
    mean, var = torch.randn(4, 5, generator=g), torch.randn(4, 5, generator=g)
    with ref_capture.capture_rng() as cap:
        zz = v.reparameterization(mean, var)
    A["reparam/mean"], A["reparam/log_var"], A["reparam/eps"], A["reparam/z"] = \
        _np(mean), _np(var), _np(cap.randn_like[0]), _np(zz)

    # TripletMultiOmicDataset index semantics (data.py:1089-1151): label -> indices map incl. "NA"
    lab = torch.tensor([0., 1., 2., 0., float("nan"), 1., 2., 0., 1., float("nan")])
    dsT = ref_capture.make_dataset(R, {"gex": torch.randn(10, 4, generator=g)}, {"c": lab}, {"c": "categorical"})
    T = R.TripletMultiOmicDataset(dsT, "c")
    A["tripletds/labels"] = _np(lab)
    A["tripletds/valid_indices"] = np.asarray(T.valid_indices, np.int64)
    for k, idx in T.label_to_indices.items():
        A[f"tripletds/idx/{k}"] = np.asarray(idx, np.int64)
    np.savez_compressed(os.path.join(out, "functions.npz"), **A)
    print(f"[golden] functions: {len(A)} arrays")


FINETUNE_FREEZES = {"enc_frozen": {"encoders": True, "supervisors": False}, "sup_frozen": {"encoders": False, "supervisors": True}}


def finetune_specs():
    return {"DirectPred": Spec("DirectPred", [("a", 33), ("b", 21)], 5, 0.4, 3, [("c", "categorical", 4), ("y", "numerical", 1)]),
            "supervised_vae": Spec("supervised_vae", [("a", 30), ("b", 18)], 6, 0.3, 3, [("y", "numerical", 1)])}


def gen_finetune_goldens(R, out):
    """The FineTuner's optimisation step (reference main.py:530-539 requires_grad flags per parameter group, :562-566 Adam over the
    trainable parameters only, :591-600 a Trainer WITHOUT gradient clipping), recorded from the reference's own model classes for
    both freeze configurations: two consecutive steps, losses, the set and values of the gradients, post-step state.  The fixture
    travels to the GPU box (the live pin in tests/test_oracle_pinning.py does not)."""
    lr, B = 3e-3, 6
    arrays = {"lr": np.array(lr), "n_steps": np.array(2)}
    for mname, spec in finetune_specs().items():
        dat, ann, vt = make_cohort(spec, 30, seed=9, missing=False)
        ds = ref_capture.make_dataset(R, dat, ann, vt)
        cfg = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": lr,
               "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": 1, "batch_size": B}
        st0 = perturbed_state(spec, seed=3)
        batches = make_batches(spec, dat, ann, B, 2, seed=4, missing=False)
        arrays[f"{mname}/spec_json"] = np.array(json.dumps(dataclasses.asdict(spec)))
        for k, v in st0.items():
            arrays[f"{mname}/state0/{k}"] = _np(v)
        for s, b in enumerate(batches):
            for i, x in enumerate(b["x"]):
                arrays[f"{mname}/batch/{s}/x/{i}"] = _np(x)
            for k, v in b["y"].items():
                arrays[f"{mname}/batch/{s}/y/{k}"] = _np(v)
        for fname, freeze in FINETUNE_FREEZES.items():
            torch.manual_seed(21)
            model = ref_capture.build_reference_model(R, spec, ds, cfg)
            model.load_state_dict(st0)
            recs = ref_capture.reference_train_steps(R, spec, model, batches, lr, clip=False, freeze=freeze)
            for s, r in enumerate(recs):
                pre = f"{mname}/{fname}/{s}"
                arrays[f"{pre}/total"] = _np(r.total).reshape(())
                arrays[f"{pre}/grad_norm"] = _np(r.grad_norm).reshape(())
                for k, v in r.draws.items():
                    arrays[f"{pre}/draws/{k}"] = _np(v)
                for k, v in r.grads.items():
                    arrays[f"{pre}/grad/{k}"] = _np(v)
                for k, v in r.state.items():
                    arrays[f"{pre}/state/{k}"] = _np(v)
            print(f"[golden] finetune_step {mname} {fname}: totals", [float(r.total) for r in recs])
    np.savez_compressed(os.path.join(out, "finetune_step.npz"), **arrays)


def main():
    if not ref_shim.available():
        sys.exit("reference not present; goldens can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    R = ref_shim.load()
    torch.set_num_threads(1)          # fixed reduction order for the recorded numbers
    only = set(sys.argv[1:])                     # optional: regenerate just the named cases
    for name, cfg in case_specs().items():
        if not only or name in only:
            gen_model_case(R, name, cfg, OUT)
    if not only or "functions" in only:
        gen_function_goldens(R, OUT)
    if not only or "finetune_step" in only:
        gen_finetune_goldens(R, OUT)


if __name__ == "__main__":
    main()
