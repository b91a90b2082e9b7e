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
