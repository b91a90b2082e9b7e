"""Loader for tests/golden/*.npz (see oracle/gen_goldens.py for how they were produced)."""
import json
import os

import numpy as np
import torch

from oracle.restate import Spec

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["directpred_2omics_multitask", "directpred_1omics_regression", "directpred_unweighted",
               "supervised_vae_2omics", "triplet_3omics", "crossmodal_2in_2out"]


def _t(a):
    return torch.from_numpy(np.array(a))


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.keys = list(self.z.keys())

    def sub(self, prefix):
        prefix = prefix.rstrip("/") + "/"
        return {k[len(prefix):]: _t(self.z[k]) for k in self.keys if k.startswith(prefix)}

    def get(self, key):
        return _t(self.z[key])

    @property
    def spec(self):
        d = json.loads(str(self.z["spec_json"]))
        d["layers"] = [tuple(x) for x in d["layers"]]
        d["variables"] = [tuple(x) for x in d["variables"]]
        return Spec(**d)

    @property
    def lr(self):
        return float(self.z["lr"])

    @property
    def n_steps(self):
        return int(self.z["n_steps"])

    def state0(self):
        return self.sub("state0")

    def batch(self, s):
        b = {"y": self.sub(f"batch/{s}/y")}
        for part in ("x", "anchor", "positive", "negative"):
            d = self.sub(f"batch/{s}/{part}")
            if d:
                b[part] = [d[str(i)] for i in range(len(d))]
        return b

    def draws(self, s):
        return self.sub(f"draws/{s}")

    def exp(self, s, what):
        return self.sub(f"exp/{s}/{what}")


class FinetuneGolden:
    """tests/golden/finetune_step.npz (oracle/gen_goldens.py: gen_finetune_goldens): the FineTuner's step recorded from the
    reference for DirectPred / supervised_vae x {encoders frozen, supervisors frozen}, clip off, two steps."""
    FREEZES = {"enc_frozen": ("encoders.",), "sup_frozen": ("MLPs.",)}

    def __init__(self, model):
        self.model = model
        self.z = np.load(os.path.join(GOLDEN_DIR, "finetune_step.npz"), allow_pickle=False)
        self.keys = list(self.z.keys())

    def sub(self, prefix):
        prefix = f"{self.model}/" + prefix.rstrip("/") + "/"
        return {k[len(prefix):]: _t(self.z[k]) for k in self.keys if k.startswith(prefix)}

    @property
    def spec(self):
        d = json.loads(str(self.z[f"{self.model}/spec_json"]))
        d["layers"] = [tuple(x) for x in d["layers"]]
        d["variables"] = [tuple(x) for x in d["variables"]]
        return Spec(**d)

    @property
    def lr(self):
        return float(self.z["lr"])

    def state0(self):
        return self.sub("state0")

    def batch(self, s):
        x = self.sub(f"batch/{s}/x")
        return {"x": [x[str(i)] for i in range(len(x))], "y": self.sub(f"batch/{s}/y")}

    def step(self, freeze, s):
        pre = f"{freeze}/{s}"
        return dict(total=_t(self.z[f"{self.model}/{pre}/total"]), grad_norm=_t(self.z[f"{self.model}/{pre}/grad_norm"]),
                    draws=self.sub(pre + "/draws"), grads=self.sub(pre + "/grad"), state=self.sub(pre + "/state"))


class FinetuneLoopGolden:
    """tests/golden/finetune_loop.npz (oracle/gen_finetune_loop_golden.py): FineTuner.run_experiments (reference main.py:575-659) driven
    over the reference's own DirectPred: 2 learning rates x 3 freeze configurations x 2 folds with early stopping, the results table, the
    best configuration and the final model continued from the last cross-validation model."""

    def __init__(self):
        self.z = np.load(os.path.join(GOLDEN_DIR, "finetune_loop.npz"), allow_pickle=False)
        self.keys = list(self.z.keys())
        d = json.loads(str(self.z["spec_json"]))
        d["layers"] = [tuple(x) for x in d["layers"]]
        d["variables"] = [tuple(x) for x in d["variables"]]
        self.spec = Spec(**d)
        self.n, self.n_splits, self.B, self.max_epoch = (int(self.z[k]) for k in ("n", "n_splits", "batch_size", "max_epoch"))
        self.kfold_seed = int(self.z["kfold_seed"])
        self.lrs = [float(x) for x in self.z["lrs"]]
        self.cfgs = json.loads(str(self.z["cfgs_json"]))
        self.results = json.loads(str(self.z["results_json"]))
        self.best = json.loads(str(self.z["best_json"]))
        self.folds = [(self.z[f"fold/{i}/train"].tolist(), self.z[f"fold/{i}/val"].tolist()) for i in range(self.n_splits)]

    def sub(self, prefix):
        prefix = prefix.rstrip("/") + "/"
        return {k[len(prefix):]: _t(self.z[k]) for k in self.keys if k.startswith(prefix)}

    @staticmethod
    def tag(unit):
        return "final" if unit == "final" else "unit/%d/%d/%d" % tuple(unit)

    def perms_fn(self, unit):
        t = self.tag(unit)
        last = max(int(k.split("/")[-1]) for k in self.keys if k.startswith(t + "/perm/"))
        return lambda e: _t(self.z[f"{t}/perm/{min(e, last)}"])         # (epochs past the stopped one are never trained on)

    def draws_fn(self, unit):
        t = self.tag(unit)
        last = max(int(k.split("/")[-1]) for k in self.keys if k.startswith(t + "/perm/"))
        # (a pipelined consumer may ask for the first batch of the epoch AFTER the one that stopped: never trained on)
        return lambda e, b: self.sub(f"{t}/draws/{min(e, last)}/{b}")

    def unit(self, unit):
        t = self.tag(unit)
        return float(self.z[t + "/val_loss"]), int(self.z[t + "/stopped_epoch"]), [float(x) for x in self.z[t + "/val_losses"]]
