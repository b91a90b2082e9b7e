"""Device-side ingest: raw omics matrices -> cleaned, imputed, (log1p-ed,) standardised fp32 cohort in HBM.

The engine-side counterpart of the matrix flow in the reference's ``DataImporter.import_data`` (reference
data.py:190-231): ``cleanup_data`` (:360-452) -> feature selection applied from outside (:351-354) -> ``harmonize``
(:503-517) -> ``transform_data`` (:519-521) -> ``normalize_data`` (:523-545) -> float32 ``[n_samples, n_features]``
tensors (:547-550).  Input is what the reference's HDF5 importer reads (``/matrix``: contiguous
``[n_samples, n_features]`` float32, h5_dataloader.py:88-116) or the float64 frame ``pd.read_csv`` yields, samples as
rows.  The matrix is copied into HBM once; every pass over it (moments, medians, sample
variances, the fused gather+impute+log1p+scale) is a HIP kernel of libfxhip.so (csrc/fx_ingest.hip).  Only the
F-length bookkeeping in between -- the variance quantile, the NaN-fraction test, feature intersection, sklearn's
"constant feature" rule -- runs on the host, in numpy fp64 on vectors of n_features elements.

What stays with the caller: reading files, sample-id / label alignment (``get_labels``, ``encode_labels``) and
Laplacian-score feature selection (``select_features``); their results are passed in as row / column index lists.
There is no CPU path: without the GPU library this module raises.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import ops
from .ops import FxError

@dataclass
class IngestResult:
    train: Dict[str, torch.Tensor]                 # {layer: fp32 [n_train_kept, n_features_kept]} in HBM
    test: Optional[Dict[str, torch.Tensor]]
    features: Dict[str, np.ndarray]                # {layer: kept columns, positions into the training input}
    test_features: Optional[Dict[str, np.ndarray]]  # same features as positions into the test input
    train_rows: np.ndarray                         # kept samples, positions into the training input
    test_rows: Optional[np.ndarray]
    scalers: Dict[str, tuple]                      # {layer: (mean_, scale_)} fp64, as StandardScaler holds them
    feature_logs: Dict[str, dict] = field(default_factory=dict)   # per split, per layer: variance / na_percent / selected


@dataclass
class _Cleaned:
    x: torch.Tensor                # raw matrix in HBM
    keep: np.ndarray               # kept feature positions (ascending)
    med: Optional[torch.Tensor]    # fp64 [F] imputation values (NaN where unused) or None when the layer has no NaN
    row_ok: np.ndarray             # bool [N]
    log: dict


class DeviceImporter:
    """``DeviceImporter(variance_threshold, na_threshold, log_transform).import_matrices(train, test)``; parameter
    names and defaults are ``DataImporter``'s (reference data.py:98-105)."""

    def __init__(self, variance_threshold: float = 0.01, na_threshold: float = 0.1, log_transform: bool = False,
                 device="cuda:0"):
        self.variance_threshold = float(variance_threshold)
        self.na_threshold = float(na_threshold)
        self.log_transform = bool(log_transform)
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise FxError("DeviceImporter needs the GPU (flexynesis_amd has no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dev = self.device              # ops.device_guard: kernels launch on this device's current stream
        self._rec = ops.ImmediateRecorder()

    # -- host -> HBM ---------------------------------------------------------------------------------------------
    def upload(self, mat) -> torch.Tensor:
        """[n_samples, n_features] float32/float64 host array -> HBM in its own dtype (one runtime copy: measured
        54-56 GB/s on the MI355X box for a 164 MB matrix, pageable or pinned alike -- scripts/bench_upload.py; a
        hand-rolled pinned double buffer was 7x slower because of its host-side memcpy).  Device tensors pass through."""
        if isinstance(mat, torch.Tensor) and mat.is_cuda:
            if mat.dtype not in (torch.float32, torch.float64) or mat.dim() != 2:
                raise FxError("upload: expected a 2-D fp32/fp64 matrix")
            return mat if mat.stride(1) == 1 else mat.contiguous()
        src = mat if isinstance(mat, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mat))
        if src.dtype not in (torch.float32, torch.float64) or src.dim() != 2:
            raise FxError(f"upload: expected a 2-D float32/float64 matrix, got {src.dtype} {tuple(src.shape)}")
        return src.contiguous().to(self.device)

    # -- cleanup_data (reference data.py:360-452) for one layer ------------------------------------------------------
    def _cleanup_layer(self, x: torch.Tensor) -> _Cleaned:
        N, F = x.shape
        f32 = x.dtype == torch.float32
        count, _, m2 = ops.col_moments(self._rec, x)
        count = count.cpu().numpy().astype(np.float64)
        m2 = m2.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            var = m2 / np.where(count > 1, count - 1.0, np.nan)          # df.var(axis=1): ddof = 1, NaN skipped
        if f32:
            var = var.astype(np.float32).astype(np.float64)              # a float32 frame holds float32 variances
        na = (N - count) / N                                             # df.isna().mean(axis=1)
        ok = var[~np.isnan(var)]
        thr = float(np.percentile(ok, self.variance_threshold * 100.0, method="linear")) if ok.size else float("nan")
        with np.errstate(invalid="ignore"):
            selected = (var >= thr) & (na < self.na_threshold)           # data.py:385-391
        keep = np.flatnonzero(selected)
        if keep.size == 0:
            raise FxError("cleanup: no feature passes the variance / NaN filters")
        keep_dev = torch.from_numpy(keep.astype(np.int32)).to(self.device)
        med = None
        nan_cols = keep[count[keep] < N]
        if nan_cols.size:                                                # data.py:402-417: NaN -> feature median
            med = torch.full((F,), float("nan"), dtype=torch.float64, device=self.device)
            ops.col_median(self._rec, x, torch.from_numpy(nan_cols.astype(np.int32)).to(self.device), med)
        rvar = ops.row_moments(self._rec, x, keep_dev, med).cpu().numpy()
        with np.errstate(invalid="ignore"):
            sd = np.sqrt(rvar)
        if f32:
            sd = sd.astype(np.float32)
        row_ok = (sd != 0) & ~np.isnan(sd)                               # data.py:424-431
        return _Cleaned(x, keep, med, row_ok, {"variance": var, "na_percent": na, "selected": selected})

    def _cleanup(self, mats: Dict[str, torch.Tensor]):
        layers = {k: self._cleanup_layer(x) for k, x in mats.items()}
        common = np.logical_and.reduce([c.row_ok for c in layers.values()])   # data.py:436-437
        return layers, np.flatnonzero(common)

    # -- StandardScaler.fit (sklearn, reference data.py:527) on the cleaned training matrix ----------------------------
    def _fit_scaler(self, c: _Cleaned, rows_dev: torch.Tensor, cols: np.ndarray):
        count, mean, m2 = ops.col_moments(self._rec, c.x, rows=rows_dev, med=c.med, log1p=self.log_transform)
        n = count.cpu().numpy().astype(np.float64)[cols]
        mean = mean.cpu().numpy()[cols]
        var = m2.cpu().numpy()[cols] / n                                 # population variance
        eps = np.finfo(np.float64).eps
        with np.errstate(invalid="ignore"):
            constant = var <= n * eps * var + (n * mean * eps) ** 2      # sklearn _is_constant_feature
            scale = np.sqrt(var)
        scale[constant] = 1.0                                            # sklearn _handle_zeros_in_scale
        return mean, scale

    def _transform(self, c: _Cleaned, rows: np.ndarray, cols: np.ndarray, mean: np.ndarray, scale: np.ndarray):
        dev = self.device
        out = torch.empty((rows.size, cols.size), dtype=torch.float32, device=dev)
        if rows.size == 0:
            return out
        ops.ingest_transform(self._rec, c.x, out, rows=torch.from_numpy(rows.astype(np.int32)).to(dev),
                             cols=torch.from_numpy(cols.astype(np.int32)).to(dev), med=c.med, log1p=self.log_transform,
                             mean=torch.from_numpy(mean).to(dev), scale=torch.from_numpy(scale).to(dev))
        return out

    # -- import_data's matrix flow (reference data.py:190-231) -----------------------------------------------------------
    @ops.device_guard
    def import_matrices(self, train: Dict[str, object], test: Optional[Dict[str, object]] = None, *,
                        selected: Optional[Dict[str, Sequence[int]]] = None,
                        train_feature_ids: Optional[Dict[str, Sequence]] = None,
                        test_feature_ids: Optional[Dict[str, Sequence]] = None,
                        train_rows: Optional[Sequence[int]] = None,
                        test_rows: Optional[Sequence[int]] = None) -> IngestResult:
        """``train`` / ``test``: {layer: [n_samples, n_features] float32|float64 array or device tensor}; the layer
        order of ``train`` is kept (it becomes ``dataset.dat.keys()``).
        ``selected``: per layer, training-feature positions chosen by the caller's feature selection; applied to the
        training side after cleanup, as ``process_data`` does.
        ``*_feature_ids``: feature names per layer when the two splits do not share a column order; ``harmonize``
        keeps the common ones in training order.  Default: column position is the feature id.
        ``*_rows``: restrict to these samples (the caller's ``get_labels`` intersection with the annotation table);
        the cleanup's sample mask is applied on top, in the given order."""
        if test is not None and set(test.keys()) != set(train.keys()):
            raise FxError("import_matrices: train and test must hold the same layers")
        tr_l, tr_rows = self._cleanup({k: self.upload(v) for k, v in train.items()})
        logs = {"train": {k: c.log for k, c in tr_l.items()}}
        feats = {k: c.keep for k, c in tr_l.items()}
        if selected:
            for k, sel in selected.items():
                kept = set(feats[k].tolist())
                feats[k] = np.asarray([int(f) for f in sel if int(f) in kept], dtype=np.int64)
        if train_rows is not None:
            ok = set(tr_rows.tolist())
            tr_rows = np.asarray([int(r) for r in train_rows if int(r) in ok], dtype=np.int64)
        te_l = te_rows = te_feats = None
        if test is not None:
            te_l, te_rows = self._cleanup({k: self.upload(test[k]) for k in train})
            logs["test"] = {k: c.log for k, c in te_l.items()}
            if test_rows is not None:
                ok = set(te_rows.tolist())
                te_rows = np.asarray([int(r) for r in test_rows if int(r) in ok], dtype=np.int64)
            te_feats = {}
            for k in train:                                              # harmonize (data.py:503-517)
                ids_tr = None if train_feature_ids is None else list(train_feature_ids[k])
                ids_te = None if test_feature_ids is None else list(test_feature_ids[k])
                if (ids_tr is None) != (ids_te is None):
                    raise FxError("import_matrices: give feature ids for both splits or for neither")
                if ids_tr is None:
                    in_test = set(te_l[k].keep.tolist())
                    common = [int(f) for f in feats[k] if int(f) in in_test]
                    feats[k] = np.asarray(common, dtype=np.int64)
                    te_feats[k] = feats[k].copy()
                else:
                    pos_te = {ids_te[int(j)]: int(j) for j in te_l[k].keep}
                    common = [int(f) for f in feats[k] if ids_tr[int(f)] in pos_te]
                    feats[k] = np.asarray(common, dtype=np.int64)
                    te_feats[k] = np.asarray([pos_te[ids_tr[f]] for f in common], dtype=np.int64)
        out_tr, out_te, scalers = {}, ({} if test is not None else None), {}
        rows_dev = torch.from_numpy(tr_rows.astype(np.int32)).to(self.device)
        for k in train:
            if feats[k].size == 0 or tr_rows.size == 0:
                raise FxError(f"import_matrices: layer {k!r} has no features or no samples left after cleanup")
            mean, scale = self._fit_scaler(tr_l[k], rows_dev, feats[k])
            scalers[k] = (mean, scale)
            out_tr[k] = self._transform(tr_l[k], tr_rows, feats[k], mean, scale)
            if test is not None:
                out_te[k] = self._transform(te_l[k], te_rows, te_feats[k], mean, scale)
        return IngestResult(out_tr, out_te, feats, te_feats, tr_rows, te_rows, scalers, logs)


def to_dataset(matrices: Dict[str, torch.Tensor], ann: Dict[str, object], variable_types: Dict[str, str],
               feature_names: Dict[str, Sequence], samples: Sequence, label_mappings: Optional[dict] = None):
    """Wrap ingested layers as the ``MultiOmicDataset`` the model classes take (reference data.py:547-566); ``dat``
    stays in HBM, so ``fit`` builds its device cohort without another copy."""
    from .data import MultiOmicDataset
    ann_t = {k: torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v for k, v in ann.items()}
    ann_t = {k: (v.float() if v.dtype in (torch.float64, torch.float32) else v) for k, v in ann_t.items()}
    return MultiOmicDataset(dict(matrices), ann_t, dict(variable_types), {k: list(v) for k, v in feature_names.items()},
                            list(samples), label_mappings or {})
