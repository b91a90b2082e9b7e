"""Generate tests/golden/ingest_*.npz FROM THE REFERENCE's own DataImporter methods (build container only).

    python -m oracle.gen_ingest_goldens

Each fixture holds seeded raw matrices ([n_samples, n_features], with NaNs, constant features, constant samples and
a test split) plus what the reference's cleanup_data -> harmonize -> transform_data -> normalize_data ->
get_torch_dataset chain (reference data.py:190-231) produced for them: float32 matrices, kept feature / sample
positions and the fitted scaler statistics.  A fixture is DATA; no reference source text is stored.
"""
from __future__ import annotations

import contextlib
import io
import os

import numpy as np

from . import ref_shim

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_case(seed: int, n_train: int, n_test: int, feats, dtype, counts: bool):
    """Seeded raw matrices with the awkward cases the reference's cleanup handles."""
    rng = np.random.default_rng(seed)
    train, test = {}, {}
    for name, F in feats:
        loc = rng.normal(size=(1, F)) * 4
        sc = rng.uniform(0.05, 3.0, size=(1, F))
        for dst, n in ((train, n_train), (test, n_test)):
            X = rng.normal(size=(n, F)) * sc + loc
            if counts:                                   # log_transform needs x > -1
                X = np.floor(np.abs(X) * 20)
            X[:, 3] = 1.0                                # constant feature
            X[:, 5] = 0.0
            X[rng.integers(0, n), 7] = np.nan            # one NaN -> imputed
            X[: max(2, n // 5), 9] = np.nan              # too many NaNs -> dropped
            nanr = rng.integers(0, n, size=F // 8)
            nanc = rng.integers(10, F, size=F // 8)
            X[nanr, nanc] = np.nan                       # scattered NaNs (even and odd non-NaN counts)
            dst[name] = X.astype(dtype)
    for dst, n in ((train, n_train), (test, n_test)):    # one uninformative sample (constant in every layer)
        for name in dst:
            dst[name][n // 2, :] = 2.0
    test[feats[0][0]][:, 11] = 5.0                       # feature constant in the test split only -> harmonize drops it
    return train, test


def reference_import(train, test, variance_threshold, na_threshold, log_transform):
    """Drive the reference's own methods on in-memory frames (features as rows, as the CSV/HDF5 readers build them)."""
    import pandas as pd
    ref_shim.install()
    from flexynesis.data import DataImporter
    di = DataImporter.__new__(DataImporter)              # no folder on disk: set what the methods read
    di.variance_threshold, di.na_threshold, di.log_transform = variance_threshold, na_threshold, log_transform
    di.feature_logs, di.scalers = {}, None

    def frames(mats):
        return {k: pd.DataFrame(np.ascontiguousarray(X.T), index=[f"f{i}" for i in range(X.shape[1])],
                                columns=[f"s{i}" for i in range(X.shape[0])]) for k, X in mats.items()}

    with contextlib.redirect_stdout(io.StringIO()):
        tr = di.cleanup_data(frames(train))
        te = di.cleanup_data(frames(test))
        tr, te = di.harmonize(tr, te)
        tr = {k: tr[k] for k in train}                   # harmonize iterates a set: restore the layer order
        te = {k: te[k] for k in train}
        if log_transform:
            tr, te = di.transform_data(tr), di.transform_data(te)
        trn = di.normalize_data(tr, scaler_type="standard", fit=True)
        ten = di.normalize_data(te, scaler_type="standard", fit=False)
    import torch
    out = {}
    for k in train:
        out[f"train/{k}"] = torch.from_numpy(np.array(trn[k].T)).float().numpy()       # data.py:549
        out[f"test/{k}"] = torch.from_numpy(np.array(ten[k].T)).float().numpy()
        out[f"features/{k}"] = np.array([int(s[1:]) for s in trn[k].index], dtype=np.int64)
        out[f"mean/{k}"] = di.scalers[k].mean_.astype(np.float64)
        out[f"scale/{k}"] = di.scalers[k].scale_.astype(np.float64)
    first = next(iter(train))
    out["train_rows"] = np.array([int(s[1:]) for s in trn[first].columns], dtype=np.int64)
    out["test_rows"] = np.array([int(s[1:]) for s in ten[first].columns], dtype=np.int64)
    return out


CASES = {
    "ingest_f64": dict(seed=11, n_train=41, n_test=17, feats=[("gex", 96), ("cnv", 64)], dtype=np.float64, counts=False,
                       log_transform=False),
    "ingest_f32": dict(seed=12, n_train=40, n_test=18, feats=[("gex", 80), ("meth", 72)], dtype=np.float32, counts=False,
                       log_transform=False),
    "ingest_f64_log": dict(seed=13, n_train=37, n_test=16, feats=[("gex", 88)], dtype=np.float64, counts=True,
                           log_transform=True),
    "ingest_f32_log": dict(seed=14, n_train=36, n_test=15, feats=[("gex", 70)], dtype=np.float32, counts=True,
                           log_transform=True),
}
VT, NAT = 0.01, 0.1      # DataImporter defaults (data.py:104-105)


def build_case(name):
    c = CASES[name]
    train, test = make_case(c["seed"], c["n_train"], c["n_test"], c["feats"], c["dtype"], c["counts"])
    exp = reference_import(train, test, VT, NAT, c["log_transform"])
    blob = {f"in_train/{k}": v for k, v in train.items()}
    blob.update({f"in_test/{k}": v for k, v in test.items()})
    blob.update({f"exp/{k}": v for k, v in exp.items()})
    blob["layers"] = np.array([k for k, _ in c["feats"]])
    blob["log_transform"] = np.array(c["log_transform"])
    blob["thresholds"] = np.array([VT, NAT])
    return blob


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in CASES:
        blob = build_case(name)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **blob)
        print(name, {k: v.shape for k, v in blob.items() if k.startswith("exp/train/")},
              "rows", blob["exp/train_rows"].size, blob["exp/test_rows"].size)


if __name__ == "__main__":
    main()
