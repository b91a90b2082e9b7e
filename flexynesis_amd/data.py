"""Dataset contract of the hot path + the device-resident cohort that feeds the engine.

``MultiOmicDataset`` / ``TripletMultiOmicDataset`` keep the reference's interface (reference
data.py:945-1151: same constructor arguments, ``__getitem__`` return structure, ``subset``, and the
triplet sampling rules incl. the "NA" label group) so that the reference's importer output and the
reference's HPO/CLI callers can use them unchanged.  ``DeviceCohort`` is what replaces the per-sample
``__getitem__`` + ``default_collate`` + per-batch H2D copy on the training path: the whole cohort
lives in HBM ([N,F] fp32 per layer, a 2048 x 40k cohort is 0.33 GB of 288 GB) and batches are
assembled by the fx_gather_rows kernel from an on-device index list.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset


class MultiOmicDataset(Dataset):
    """In-memory multi-omic dataset (interface of reference data.py:945-1085)."""

    def __init__(self, dat, ann, variable_types, features, samples, label_mappings, feature_ann=None):
        self.dat = dat
        self.ann = ann
        self.variable_types = variable_types
        self.features = features
        self.samples = samples
        self.label_mappings = label_mappings
        self.feature_ann = feature_ann or {}

    def __getitem__(self, index):
        return ({k: v[index] for k, v in self.dat.items()},
                {k: v[index] for k, v in self.ann.items()},
                self.samples[index])

    def __len__(self):
        return len(self.samples)

    def subset(self, indices):
        return MultiOmicDataset({k: v[indices] for k, v in self.dat.items()},
                                {k: v[indices] for k, v in self.ann.items()},
                                self.variable_types, self.features, [self.samples[i] for i in indices],
                                self.label_mappings, self.feature_ann)

    def get_feature_subset(self, feature_df):
        import pandas as pd
        wanted = feature_df.groupby("layer")["name"].apply(list).to_dict()
        frames = []
        for layer, names in wanted.items():
            if layer not in self.dat:
                print(f"Layer {layer} not found in the dataset.")
                continue
            pos = {f: i for i, f in enumerate(self.features[layer])}
            keep = [f for f in names if f in pos]
            block = self.dat[layer][:, [pos[f] for f in keep]]
            frames.append(pd.DataFrame(np.asarray(block), columns=[f"{layer}_{f}" for f in keep]))
        out = pd.concat(frames, axis=1)
        out.index = self.samples
        return out

    def get_dataset_stats(self):
        stats = {": ".join(["feature_count in", k]): v.shape[1] for k, v in self.dat.items()}
        stats["sample_count"] = len(self.samples)
        return stats


def triplet_label_index(labels: np.ndarray):
    """label -> sample indices, with all NaN labels forming one extra group "NA"
    (reference data.py:1133-1151)."""
    valid = [l for l in labels if not np.isnan(l)]
    labels_set = set(valid)
    label_to_indices = {l: np.where(labels == l)[0] for l in labels_set}
    na = np.where(np.isnan(labels))[0]
    if len(na) > 0:
        labels_set.add("NA")
        label_to_indices["NA"] = na
    return labels_set, label_to_indices


class TripletMultiOmicDataset(Dataset):
    """(anchor, positive, negative, y) sampler by ``main_var`` (interface of reference data.py:1089-1151).
    Anchors are the samples with a non-NaN main label; the positive is a *different* sample of the same
    label; the negative comes from a uniformly chosen other label group (the "NA" group included)."""

    def __init__(self, mydataset, main_var):
        self.dataset = mydataset
        self.main_var = main_var
        labels = np.asarray(self.dataset.ann[self.main_var])
        self.labels_set, self.label_to_indices = triplet_label_index(labels)
        self.valid_indices = [i for i, l in enumerate(labels) if not np.isnan(l)]

    def sample_indices(self, index):
        real = self.valid_indices[index]
        label = float(np.asarray(self.dataset.ann[self.main_var][real]))
        pool = self.label_to_indices[label]
        if len(pool) < 2:
            raise ValueError(f"label {label!r} of '{self.main_var}' has a single member: no positive exists")
        pos = real
        while pos == real:
            pos = int(np.random.choice(pool))
        neg_label = random.choice(list(self.labels_set - {label}))
        neg = int(np.random.choice(self.label_to_indices[neg_label]))
        return real, pos, neg

    def __getitem__(self, index):
        real, pos, neg = self.sample_indices(index)
        anchor, y_dict, _ = self.dataset[real]
        return anchor, self.dataset[pos][0], self.dataset[neg][0], y_dict

    def __len__(self):
        return len(self.valid_indices)


class DeviceCohort:
    """All layers and labels of a cohort resident in HBM as contiguous fp32 tensors."""

    def __init__(self, dat: Dict[str, torch.Tensor], ann: Dict[str, torch.Tensor], device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceCohort lives in GPU memory (no CPU path)")
        self.layers = list(dat.keys())                 # order = dataset.dat.keys() (reference direct_pred.py:68)
        self.dat = {k: torch.as_tensor(v).to(self.device, torch.float32).contiguous() for k, v in dat.items()}
        self.ann = {k: torch.as_tensor(v).to(self.device, torch.float32).contiguous() for k, v in ann.items()}
        self.n = next(iter(self.dat.values())).shape[0]

    @classmethod
    def from_dataset(cls, ds, device):
        return cls(ds.dat, ds.ann, device)

    def nbytes(self) -> int:
        return sum(t.numel() * 4 for t in list(self.dat.values()) + list(self.ann.values()))


def synthetic_cohort(layers, n: int, device, seed: int = 1234, n_classes: int = 4) -> DeviceCohort:
    """Seeded synthetic cohort of SURVEY.md section 8(d), generated directly in HBM: N(0,1) features
    (what StandardScaler output looks like, reference data.py:531), a regression target driven by the first
    16 features of the first layer, a balanced categorical label, tie-free survival times and events."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    dat = {name: torch.randn(n, F, generator=g, device=dev) for name, F in layers}
    first = dat[layers[0][0]]
    k = min(16, first.shape[1])
    ann = {
        "y": first[:, :k].sum(1) / 4 + 0.1 * torch.randn(n, generator=g, device=dev),
        "c": torch.randint(0, n_classes, (n,), generator=g, device=dev).float(),
        "time": torch.rand(n, generator=g, device=dev) * 10,
        "event": (torch.rand(n, generator=g, device=dev) < 0.5).float(),
    }
    return DeviceCohort(dat, ann, dev)
