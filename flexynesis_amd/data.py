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
        # Rows are stored with a pitch of a multiple of 4 floats (the pad columns are zero): the batch assembly reads 16 bytes per
        # lane whatever the layer's feature count is (engine.StepPlan gathers from ``source``); ``dat`` are the [N, F] views.
        self._rows, self.dat = {}, {}
        for k, v in dat.items():
            v = torch.as_tensor(v)
            F = int(v.shape[1])
            Fp = (F + 3) // 4 * 4
            if Fp == F:
                buf = v.to(self.device, torch.float32).contiguous()
            else:
                buf = torch.zeros(v.shape[0], Fp, dtype=torch.float32, device=self.device)
                buf[:, :F].copy_(v)
            self._rows[k], self.dat[k] = buf, buf[:, :F]
        self.ann = {k: torch.as_tensor(v).to(self.device, torch.float32).contiguous() for k, v in ann.items()}
        self.n = next(iter(self.dat.values())).shape[0]

    def source(self, name: str, width: int) -> torch.Tensor:
        """Layer ``name`` as a [N, width] tensor for the batch assembly: width = its feature count, or that rounded up to 4 (zero columns)."""
        buf = self._rows[name]
        if width == buf.shape[1]:
            return buf
        if width == self.dat[name].shape[1]:
            return self.dat[name]
        raise ValueError(f"layer {name!r}: {self.dat[name].shape[1]} features, asked for a width of {width}")

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


class MultiOmicDatasetNW(Dataset):
    """Network view of a MultiOmicDataset for the GNN model (reference data.py:1153-1266): nodes = the features that
    occur both in some omics layer and in the interaction table, one feature per layer on every node (sorted layer
    names), features a layer lacks filled with the per-sample median over the nodes, one ``edge_index`` for all
    samples.  ``interaction_df`` has columns protein1 / protein2 (the reference's STRING table); an ``edge_index``
    [2, E] over ``node_names`` can be given instead.
    ``dat`` / ``features`` expose the node tensor as the single pseudo-layer "nodes" ([n_samples, nodes * layers]) so the
    engine's cohort and fit loop see an ordinary dataset."""

    def __init__(self, multiomic_dataset, interaction_df=None, modality_order=None, *, edge_index=None, node_names=None):
        self.multiomic_dataset = multiomic_dataset
        self.interaction_df = interaction_df
        md = multiomic_dataset
        self.modality_order = modality_order if modality_order else sorted(md.dat.keys())
        feats = {k: [str(f) for f in md.features[k]] for k in md.dat.keys()}
        all_feats = set().union(*(set(v) for v in feats.values()))
        if interaction_df is not None:
            p1, p2 = [str(a) for a in interaction_df["protein1"]], [str(a) for a in interaction_df["protein2"]]
            self.common_features = sorted(all_feats & (set(p1) | set(p2)))
            self.gene_to_index = {g: i for i, g in enumerate(self.common_features)}
            keep = [(self.gene_to_index[a], self.gene_to_index[b]) for a, b in zip(p1, p2)
                    if a in self.gene_to_index and b in self.gene_to_index]
            self.edge_index = torch.tensor(keep, dtype=torch.long).t().reshape(2, -1)
        else:
            if edge_index is None or node_names is None:
                raise ValueError("give interaction_df, or edge_index together with node_names")
            self.common_features = [str(n) for n in node_names]
            self.gene_to_index = {g: i for i, g in enumerate(self.common_features)}
            self.edge_index = torch.as_tensor(edge_index, dtype=torch.long).reshape(2, -1)
        self.samples = md.samples
        self.variable_types = md.variable_types
        self.label_mappings = md.label_mappings
        self.ann = md.ann
        self.labels = dict(md.ann)
        self.node_features_tensor = self.precompute_node_features()

    def precompute_node_features(self):
        md = self.multiomic_dataset
        first = next(iter(md.dat.values()))
        n, nodes, dev = len(self.samples), len(self.common_features), first.device
        order = sorted(md.dat.keys())                                   # reference data.py:1219
        out = torch.full((n, nodes, len(order)), float("nan"), dtype=torch.float32, device=dev)
        for i, layer in enumerate(order):
            pos = {str(f): j for j, f in enumerate(md.features[layer])}
            have = [(self.gene_to_index[gname], pos[gname]) for gname in self.common_features if gname in pos]
            if have:
                node_pos = torch.tensor([a for a, _ in have], dtype=torch.long, device=dev)
                col = torch.tensor([b for _, b in have], dtype=torch.long, device=dev)
                out[:, node_pos, i] = md.dat[layer][:, col].to(torch.float32)
        med = torch.nanmedian(out, dim=1, keepdim=True).values           # per sample and layer, over the nodes
        nan = torch.isnan(out)
        out[nan] = med.expand_as(out)[nan]
        return out

    @property
    def dat(self):
        return {"nodes": self.node_features_tensor.reshape(len(self.samples), -1)}

    @property
    def features(self):
        k = self.node_features_tensor.shape[2]
        return {"nodes": [f"{g}:{j}" for g in self.common_features for j in range(k)]}

    def subset(self, indices):
        sub = self.multiomic_dataset.subset(indices)
        return MultiOmicDatasetNW(sub, self.interaction_df, self.modality_order,
                                  edge_index=None if self.interaction_df is not None else self.edge_index,
                                  node_names=None if self.interaction_df is not None else self.common_features)

    def __getitem__(self, idx):
        y = {k: v[idx] for k, v in self.labels.items()}
        return self.node_features_tensor[idx], y, self.samples[idx]

    def __len__(self):
        return len(self.samples)
