"""Generate tests/golden/finetune_loop.npz FROM THE REFERENCE's model class (build container only):

    python -m oracle.gen_finetune_loop_golden

``FineTuner.run_experiments`` (reference main.py:575-659) driven by hand over the reference's own DirectPred -- Lightning is absent
from this image, so the Trainer's part of the loop follows oracle/loop.py's restatement ([L] points there) while everything the
reference's code does inside it is the reference's: ``copy.deepcopy`` of the original model per fit, ``apply_freeze_config``'s
requires_grad flags, ``Adam(filter(requires_grad), lr)``, ``training_step(log=False)`` -> backward -> step WITHOUT clipping,
``validation_step`` over the fold's validation rows, EarlyStopping(patience 3), the mean over the folds, the best configuration, and
the final model continuing from the LAST cross-validation model on all samples for the best configuration's mean stopped epoch.
Recorded: inputs (spec, initial state, cohort, folds, every shuffle and dropout mask) and what came out (per fit: validation loss of
every epoch, stopped epoch; the results table; the final state)."""
from __future__ import annotations

import copy
import dataclasses
import json
import math
import os

import numpy as np
import torch

from . import ref_capture, ref_shim
from .gen_goldens import make_cohort, perturbed_state
from .restate import Spec

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SPEC = Spec("DirectPred", [("gex", 36), ("cnv", 24)], 6, 0.3, 4, [("y", "numerical", 1), ("c", "categorical", 3)])
N, N_SPLITS, B, MAX_EPOCH, KSEED = 60, 2, 10, 9, int(os.environ.get("FT_KSEED", "4"))
LRS = [float(x) for x in os.environ.get("FT_LRS", "1e-2,2e-3").split(",")]
CSEED, SSEED = int(os.environ.get("FT_CSEED", "51")), int(os.environ.get("FT_SSEED", "17"))
CFGS = [{"encoders": True, "supervisors": False}, {"encoders": False, "supervisors": True}, {"encoders": False, "supervisors": False}]


def kfold(n, k, seed):
    """flexynesis_amd.fit.kfold_indices stated again (sklearn KFold(shuffle=True) sizes; the test checks they agree)."""
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(int(seed))).tolist()
    sizes = [n // k + (1 if i < n % k else 0) for i in range(k)]
    folds, o = [], 0
    for sz in sizes:
        val = sorted(perm[o:o + sz])
        held = set(val)
        folds.append(([i for i in range(n) if i not in held], val))
        o += sz
    return folds


def _stop_of(vals, patience=3):
    best, wait = math.inf, 0
    for e, cur in enumerate(vals):
        if cur < best:
            best, wait = cur, 0
        else:
            wait += 1
            if wait >= patience:
                return e
    return 0


def robust(vals, stopped, rel=1.5e-3, trials=400):
    """Would validation losses that are off by ~rel (what two implementations differ by here: the noise walk of the zero-gradient
    biases) have led to the same early-stopping outcome?"""
    rng = np.random.default_rng(0)
    v = np.asarray(vals)
    for _ in range(trials):
        p = v * (1.0 + rel * rng.uniform(-1, 1, v.shape))
        # a run that stopped would not have produced the later epochs; one that did not stop is judged on what it produced
        if _stop_of(list(p)) != stopped and not (stopped and _stop_of(list(p)) == 0 and len(v) < MAX_EPOCH and False):
            return False
    return True


def main():
    for seed in range(1234, 1334):
        if generate(seed):
            return
    raise SystemExit("no seed gave decisive early-stopping margins")


def generate(seed):
    R = ref_shim.load()
    torch.set_num_threads(1)
    torch.manual_seed(seed)            # the reference's dropout draws come from torch's global generator
    all_robust = True
    spec = SPEC
    dat, ann, vt = make_cohort(spec, N, seed=CSEED, missing=False)
    ds = ref_capture.make_dataset(R, dat, ann, vt)
    cfg = {"latent_dim": spec.latent_dim, "hidden_dim_factor": spec.hidden_dim_factor, "lr": LRS[0],
           "supervisor_hidden_dim": spec.supervisor_hidden_dim, "epochs": MAX_EPOCH, "batch_size": B}
    original = ref_capture.build_reference_model(R, spec, ds, cfg)
    st0 = perturbed_state(spec, seed=SSEED)
    original.load_state_dict(st0)
    folds = kfold(N, N_SPLITS, KSEED)
    keys = [v[0] for v in spec.variables]
    g = torch.Generator().manual_seed(99)
    out = {"torch_seed": seed, "spec_json": json.dumps(dataclasses.asdict(spec)), "n": N, "n_splits": N_SPLITS, "batch_size": B, "max_epoch": MAX_EPOCH,
           "kfold_seed": KSEED, "lrs": np.asarray(LRS), "cfgs_json": json.dumps(CFGS)}
    for k, v in st0.items():
        out[f"state0/{k}"] = v.numpy()
    for k, v in dat.items():
        out[f"dat/{k}"] = v.numpy()
    for k, v in ann.items():
        out[f"ann/{k}"] = v.numpy()
    for fi, (tr, va) in enumerate(folds):
        out[f"fold/{fi}/train"], out[f"fold/{fi}/val"] = np.asarray(tr, np.int64), np.asarray(va, np.int64)

    def batch(rows):
        return ref_capture.reference_batch(spec, {"x": [dat[nm][rows] for nm, _ in spec.layers], "y": {k: ann[k][rows] for k in keys}})

    def apply_freeze(model, c):                 # FineTuner.apply_freeze_config (main.py:530-539), on the reference's modules
        for enc in model.encoders:
            for p in enc.parameters():
                p.requires_grad = not c["encoders"]
        for mlp in model.MLPs.values():
            for p in mlp.parameters():
                p.requires_grad = not c["supervisors"]

    def train_epochs(model, lr, rows_all, tag, epochs, val_rows=None, patience=0):
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=lr)     # main.py:562-566
        best, wait, stopped, vals = math.inf, 0, 0, []
        rows_all = torch.as_tensor(rows_all, dtype=torch.int64)
        for e in range(epochs):
            perm = torch.randperm(rows_all.numel(), generator=g)
            out[f"{tag}/perm/{e}"] = perm.numpy()
            model.train()
            order = rows_all[perm]
            for b, s in enumerate(range(0, order.numel(), B)):
                rows = order[s:s + B]
                opt.zero_grad()
                with ref_capture.capture_rng() as cap:
                    loss = model.training_step(batch(rows), b, log=False)
                loss.backward()
                opt.step()                                                                   # (no gradient clipping: main.py:591-600)
                for k, v in ref_capture.name_draws(spec, cap).items():
                    out[f"{tag}/draws/{e}/{b}/{k}"] = v.numpy()
            if val_rows is None:
                continue
            model.eval()
            tot, cnt = 0.0, 0
            vr = torch.as_tensor(val_rows, dtype=torch.int64)
            with torch.no_grad():
                for bi, s in enumerate(range(0, vr.numel(), B)):
                    rows = vr[s:s + B]
                    vl = model.validation_step(batch(rows), bi, log=False)
                    tot += float(torch.as_tensor(vl).reshape(-1)[0]) * rows.numel()
                    cnt += rows.numel()
            cur = tot / cnt
            vals.append(cur)
            if patience:
                if not math.isfinite(cur):
                    stopped = e
                    break
                if cur < best:
                    best, wait = cur, 0
                else:
                    wait += 1
                    if wait >= patience:
                        stopped = e
                        break
        return vals, stopped

    results, last = [], None
    for li, lr in enumerate(LRS):
        for ci, c in enumerate(CFGS):
            fl, fe = [], []
            for fi, (tr, va) in enumerate(folds):
                tag = f"unit/{li}/{ci}/{fi}"
                model = copy.deepcopy(original)
                apply_freeze(model, c)
                vals, stopped = train_epochs(model, lr, tr, tag, MAX_EPOCH, va, patience=3)
                # how decisive were the early-stopping comparisons?  (smallest relative gap between an epoch's loss and the best so far)
                bs, gaps = math.inf, []
                for v_ in vals:
                    if bs < math.inf:
                        gaps.append(abs(v_ - bs) / bs)
                    bs = min(bs, v_)
                ok = robust(vals, stopped)
                all_robust = all_robust and ok
                print(f"  {tag}: stopped {stopped}, min decision gap {min(gaps):.2e}, robust {ok}, val {[round(v_, 4) for v_ in vals]}")
                out[f"{tag}/val_losses"] = np.asarray(vals, np.float64)
                out[f"{tag}/stopped_epoch"] = np.int64(stopped)
                out[f"{tag}/val_loss"] = np.float64(vals[-1])          # trainer.validate after fit: the same weights, the same rows
                fl.append(vals[-1])
                fe.append(stopped)
                last = model
            rec = {"learning_rate": lr, "average_val_loss": float(np.mean(fl)), "freeze": c, "epochs": int(np.mean(fe))}
            results.append(rec)
            print("[finetune golden]", rec, "stopped", fe)
    # The trajectories are chaotic at the level that matters for early stopping: the biases in front of a BatchNorm have a true gradient
    # of 0, every implementation computes its own rounding noise for them, Adam turns it into +-lr steps and eval-mode BatchNorm sees
    # them.  A golden is only useful if other implementations reach the same DECISIONS: replay every fit with
    # the restatement from three differently perturbed copies of those biases and keep this seed only if all of them stop where the
    # reference stopped, with validation losses within 1 %.
    if all_robust:
        from . import loop as L
        stt = {k: v.clone() for k, v in st0.items()}
        for li, lr in enumerate(LRS):
            for ci, c in enumerate(CFGS):
                frozen = tuple(L.FREEZE_PREFIXES[k] for k in ("encoders", "supervisors") if c.get(k))
                for fi, (tr, va) in enumerate(folds):
                    tag = f"unit/{li}/{ci}/{fi}"
                    last_e = max(int(k.split("/")[-1]) for k in out if k.startswith(tag + "/perm/"))
                    pf = lambda e, tag=tag, last_e=last_e: torch.from_numpy(out[f"{tag}/perm/{min(e, last_e)}"])
                    df = lambda e, b, tag=tag, last_e=last_e: {k[len(f"{tag}/draws/{min(e, last_e)}/{b}/"):]: torch.from_numpy(v) for k, v in out.items()
                                                                if k.startswith(f"{tag}/draws/{min(e, last_e)}/{b}/")}
                    for trial in range(3):
                        gp = torch.Generator().manual_seed(1000 + trial)
                        pert = {k: (v + 3e-3 * torch.randn(v.shape, generator=gp) if (k.endswith("layer_1.bias") or k.endswith("layer_out.bias")
                                                                                     or k == "fusion_block.bias") else v.clone())
                                for k, v in stt.items()}
                        r = L.fit_reference(spec, pert, dat, ann, tr, va, batch_size=B, epochs=MAX_EPOCH, lr=float(lr), patience=3,
                                            perms=[pf(e) for e in range(MAX_EPOCH)], draws_fn=df, clip=False, frozen=frozen, drop_last=False)
                        ref_stop, ref_val = int(out[f"{tag}/stopped_epoch"]), float(out[f"{tag}/val_loss"])
                        if r["stopped_epoch"] != ref_stop or abs(r["val_loss"] - ref_val) > 1e-2 * ref_val:
                            print(f"  {tag}: a perturbed replay stops at {r['stopped_epoch']} (reference {ref_stop}), val {r['val_loss']:.4f} vs {ref_val:.4f}: seed rejected")
                            all_robust = False
                            break
                    if not all_robust:
                        break
                if not all_robust:
                    break
            if not all_robust:
                break
    best = min(results, key=lambda r: r["average_val_loss"])
    print("[finetune golden] seed", seed, "best", best, "all early-stopping outcomes robust:", all_robust)
    ranked = sorted(r["average_val_loss"] for r in results)
    if not all_robust or best["epochs"] < 1 or (ranked[1] - ranked[0]) / ranked[0] < 3e-3:
        return False                      # (also: the best configuration must win by more than the implementations differ)
    final = copy.deepcopy(last)
    apply_freeze(final, best["freeze"])
    if best["epochs"] > 0:
        train_epochs(final, best["learning_rate"], list(range(N)), "final", best["epochs"])
    out["results_json"] = json.dumps(results)
    out["best_json"] = json.dumps(best)
    for k, v in last.state_dict().items():
        out[f"state_last/{k}"] = v.detach().numpy().copy()
    for k, v in final.state_dict().items():
        out[f"state_final/{k}"] = v.detach().numpy().copy()
    path = os.path.join(GOLDEN_DIR, "finetune_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    return True


if __name__ == "__main__":
    main()
