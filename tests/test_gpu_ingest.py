"""Device ingest (csrc/fx_ingest.hip through the C ABI, flexynesis_amd/ingest.py) against the pinned CPU restatement
(oracle/ingest_restate.py) and the reference-generated fixtures (tests/golden/ingest_*.npz).

Bar: kept feature / sample sets identical; fp32 outputs equal to the reference's to within 1 ulp with >= 99.99 %
bit-identical (the fp64 statistics differ from pandas / sklearn only by summation order, ~1e-16 relative, which can
flip the last fp32 rounding of an output once in ~1e7 elements); with log_transform the fp32-frame path may differ by
one more ulp before scaling because numpy's float32 log1p is not correctly rounded (tolerance stated in the test)."""
import numpy as np
import pytest
import torch

from oracle import ingest_restate as R
from test_ingest_pinning import CASES, load_case

pytestmark = pytest.mark.gpu


def _imp(**kw):
    from flexynesis_amd.ingest import DeviceImporter
    return DeviceImporter(device="cuda:0", **kw)


def ulps(a, b):
    """distance in fp32 units in the last place"""
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def close_fp32(got, exp, max_ulp=1, min_exact=0.9999, what=""):
    got = got.cpu().numpy() if torch.is_tensor(got) else got
    assert got.dtype == np.float32 and got.shape == exp.shape, (what, got.shape, exp.shape)
    u = ulps(got, exp)
    # near zero an ulp is tiny: also accept an absolute 1e-7 there (values are O(1) standardised data)
    bad = (u > max_ulp) & (np.abs(got - exp) > 1e-7)
    assert not bad.any(), (what, int(bad.sum()), int(u.max()), float(np.abs(got - exp).max()))
    assert (got == exp).mean() >= min_exact, (what, float((got == exp).mean()))


@pytest.mark.parametrize("case", CASES)
def test_import_matches_reference_fixture(case):
    layers, train, test, exp, vt, nat, logt = load_case(case)
    res = _imp(variance_threshold=vt, na_threshold=nat, log_transform=logt).import_matrices(train, test)
    assert list(res.train.keys()) == layers
    assert np.array_equal(res.train_rows, exp["train_rows"]) and np.array_equal(res.test_rows, exp["test_rows"])
    f32log = logt and train[layers[0]].dtype == np.float32
    for k in layers:
        assert np.array_equal(res.features[k], exp[f"features/{k}"]), k
        st = 1e-6 if f32log else 1e-12
        np.testing.assert_allclose(res.scalers[k][0], exp[f"mean/{k}"], rtol=st, atol=st)
        np.testing.assert_allclose(res.scalers[k][1], exp[f"scale/{k}"], rtol=st)
        for side, got in (("train", res.train[k]), ("test", res.test[k])):
            assert got.is_cuda and got.dtype == torch.float32
            if f32log:      # float32 np.log1p: 1 ulp of log1p(x) (<= 16 here) over a scale >= 0.05 -> a few 1e-5
                np.testing.assert_allclose(got.cpu().numpy(), exp[f"{side}/{k}"], rtol=0, atol=4e-5)
            else:
                close_fp32(got, exp[f"{side}/{k}"], what=(case, k, side), min_exact=0.999)


@pytest.mark.parametrize("seed,dtype,N,F", [(1, np.float64, 300, 1500), (2, np.float32, 257, 1000), (3, np.float64, 64, 4097),
                                            (4, np.float32, 1000, 130)])
def test_import_matches_oracle_seeded(seed, dtype, N, F):
    rng = np.random.default_rng(seed)
    def mk(n):
        X = rng.normal(size=(n, F)) * rng.uniform(0.01, 5, size=(1, F)) + rng.normal(size=(1, F)) * 30
        X[rng.integers(0, n, F // 4), rng.integers(0, F, F // 4)] = np.nan
        X[: n // 3, 17] = np.nan
        X[:, 40:44] = 3.0
        X[n // 3, :] = 0.0
        X[n - 1, :] = -1.5
        return X.astype(dtype)
    train = {"gex": mk(N), "cnv": mk(N)[:, : F // 2].copy()}
    test = {"gex": mk(N // 3 + 2), "cnv": mk(N // 3 + 2)[:, : F // 2].copy()}
    out = R.import_matrices(train, test, variance_threshold=0.05, na_threshold=0.15)
    res = _imp(variance_threshold=0.05, na_threshold=0.15).import_matrices(train, test)
    assert np.array_equal(res.train_rows, out["train_rows"]) and np.array_equal(res.test_rows, out["test_rows"])
    assert N // 3 not in res.train_rows and N - 1 not in res.train_rows
    for k in train:
        assert np.array_equal(res.features[k], out["features"][k]), k
        assert 17 not in res.features[k] and 41 not in res.features[k]
        close_fp32(res.train[k], out["train"][k], what=(k, "train"))
        close_fp32(res.test[k], out["test"][k], what=(k, "test"))


def test_col_moments_and_median_kernels():
    from flexynesis_amd import ops
    rec = ops.ImmediateRecorder()
    rng = np.random.default_rng(9)
    for dtype, N, F in ((np.float32, 777, 601), (np.float64, 130, 2050), (np.float32, 5, 33), (np.float64, 1, 7),
                        (np.float32, 4500, 24), (np.float64, 9000, 9)):      # > 4096 rows: past the median kernel's register cache
        X = (rng.normal(size=(N, F)) * 3 + 1e4).astype(dtype)         # mean >> std: the shifted sums must not cancel
        X[rng.integers(0, N, N), rng.integers(0, F, N)] = np.nan
        X[:, 2] = np.nan                                                # all-NaN column
        if N > 4:
            X[0:3, 5] = [np.inf, -np.inf, 0.0][: min(3, N)]
            X[:, 6] = np.where(np.arange(N) % 2 == 0, -0.0, 0.0)
        x = torch.from_numpy(X).cuda()
        count, mean, m2 = ops.col_moments(rec, x)
        cnt = (~np.isnan(X)).sum(0)
        assert np.array_equal(count.cpu().numpy(), cnt)
        Xd = X.astype(np.float64)
        ok = cnt > 0
        finite = np.isfinite(np.where(np.isnan(Xd), 0, Xd)).all(0)
        with np.errstate(all="ignore"):
            ref_mean = np.nanmean(Xd, axis=0)
            ref_m2 = np.nansum((Xd - ref_mean) ** 2, axis=0)
        sel = ok & finite
        np.testing.assert_allclose(mean.cpu().numpy()[sel], ref_mean[sel], rtol=1e-14, atol=1e-300)
        np.testing.assert_allclose(m2.cpu().numpy()[sel], ref_m2[sel], rtol=1e-11, atol=1e-20)
        assert np.isnan(mean.cpu().numpy()[2]) and count.cpu().numpy()[2] == 0
        # rows subset + imputation + log1p
        rows = np.sort(rng.choice(N, size=max(1, N // 2), replace=False)).astype(np.int32)
        med = torch.full((F,), float("nan"), dtype=torch.float64, device="cuda")
        cols = torch.arange(F, dtype=torch.int32, device="cuda")
        ops.col_median(rec, x, cols, med)
        with np.errstate(all="ignore"):
            ref_med = np.array([np.nanmedian(X[:, c]) if cnt[c] else np.nan for c in range(F)]).astype(np.float64)
        assert np.array_equal(med.cpu().numpy(), ref_med, equal_nan=True), dtype
        Xa = np.abs(np.where(np.isnan(X), ref_med.astype(dtype)[None, :], X))
        xa = torch.from_numpy(Xa.astype(dtype)).cuda()
        c2, mean2, m22 = ops.col_moments(rec, xa, rows=torch.from_numpy(rows).cuda(), log1p=True)
        with np.errstate(all="ignore"):
            L = np.log1p(Xa.astype(dtype))[rows].astype(np.float64)
            rm = np.nanmean(L, axis=0)
        sel = np.isfinite(rm)
        np.testing.assert_allclose(mean2.cpu().numpy()[sel], rm[sel], rtol=1e-6 if dtype == np.float32 else 1e-13)


def test_row_moments_and_transform_kernels():
    from flexynesis_amd import ops
    rec = ops.ImmediateRecorder()
    rng = np.random.default_rng(10)
    N, F = 70, 900
    X = rng.normal(size=(N, F)) * 2 + 5
    X[rng.integers(0, N, 200), rng.integers(0, F, 200)] = np.nan
    X[11, :] = 4.0
    med_np = np.nanmedian(X, axis=0)
    cols = np.sort(rng.choice(F, size=500, replace=False)).astype(np.int32)
    x = torch.from_numpy(X).cuda()
    med = torch.from_numpy(med_np).cuda()
    var = ops.row_moments(rec, x, torch.from_numpy(cols).cuda(), med).cpu().numpy()
    Xi = np.where(np.isnan(X), med_np[None, :], X)[:, cols]
    np.testing.assert_allclose(var, Xi.var(axis=1, ddof=1), rtol=1e-12, atol=1e-18)
    assert var[11] == 0.0
    rows = np.array([5, 3, 69, 0, 11], dtype=np.int32)                   # arbitrary order, as get_labels yields
    mean, scale = Xi.mean(0), Xi.std(0)
    out = torch.empty((rows.size, cols.size), dtype=torch.float32, device="cuda")
    ops.ingest_transform(rec, x, out, rows=torch.from_numpy(rows).cuda(), cols=torch.from_numpy(cols).cuda(), med=med,
                         mean=torch.from_numpy(mean).cuda(), scale=torch.from_numpy(scale).cuda())
    exp = ((Xi[rows] - mean) / scale).astype(np.float32)
    assert np.array_equal(out.cpu().numpy(), exp)
    raw = torch.empty((N, F), dtype=torch.float32, device="cuda")        # no rows / cols / scaling: plain cast + impute
    ops.ingest_transform(rec, x, raw, med=med)
    assert np.array_equal(raw.cpu().numpy(), np.where(np.isnan(X), med_np[None, :], X).astype(np.float32))


def test_feature_ids_selection_and_row_lists():
    rng = np.random.default_rng(21)
    N, F = 90, 300
    Xtr = rng.normal(size=(N, F)) * rng.uniform(0.5, 2, size=(1, F))
    Xte_full = rng.normal(size=(40, F)) * rng.uniform(0.5, 2, size=(1, F))
    perm = rng.permutation(F)[:250]                                         # the test split holds other columns, permuted
    ids_tr = [f"g{i}" for i in range(F)]
    ids_te = [f"g{i}" for i in perm]
    sel = np.sort(rng.choice(F, size=120, replace=False))
    rows = rng.permutation(N)[:60]
    res = _imp(variance_threshold=0.0).import_matrices({"gex": Xtr}, {"gex": Xte_full[:, perm].copy()}, selected={"gex": sel},
                                                        train_feature_ids={"gex": ids_tr}, test_feature_ids={"gex": ids_te},
                                                        train_rows=rows)
    common = [int(f) for f in sel if f in set(perm.tolist())]
    assert res.features["gex"].tolist() == common
    assert [ids_te[j] for j in res.test_features["gex"]] == [ids_tr[f] for f in common]
    assert res.train_rows.tolist() == rows.tolist()
    mean, scale = R.scaler_fit(Xtr[rows][:, common])
    close_fp32(res.train["gex"], R.scaler_transform(Xtr[rows][:, common], mean, scale), what="train")
    close_fp32(res.test["gex"], R.scaler_transform(Xte_full[:, common], mean, scale), what="test")


def test_full_size_properties_and_fit_on_ingested_cohort():
    """cfg2-sized layer (2048 x 20000 fp32): output columns are standardised, the ingest is deterministic, ingesting
    already-standardised data again is (numerically) the identity, and the device-resident result trains."""
    g = torch.Generator(device="cuda").manual_seed(3)
    N, F = 2048, 20000
    raw = torch.randn(N, F, generator=g, device="cuda") * (torch.rand(1, F, generator=g, device="cuda") * 4 + 0.1) + 7
    raw[torch.randint(0, N, (5000,), generator=g, device="cuda"), torch.randint(0, F, (5000,), generator=g, device="cuda")] = float("nan")
    imp = _imp(variance_threshold=0.01, na_threshold=0.1)
    res = imp.import_matrices({"gex": raw})
    out = res.train["gex"]
    assert out.shape[0] == N and 0.985 * F <= out.shape[1] <= 0.995 * F and bool(torch.isfinite(out).all())
    m = out.double().mean(0)
    s = out.double().std(0, unbiased=False)
    assert float(m.abs().max()) < 1e-6 and float((s - 1).abs().max()) < 1e-5
    res2 = imp.import_matrices({"gex": raw})
    assert torch.equal(res2.train["gex"], out)                                     # bit-reproducible
    again = _imp(variance_threshold=0.0).import_matrices({"gex": out}).train["gex"]
    assert float((again - out).abs().max()) < 2e-6                                  # idempotent on standardised data
    # the HBM-resident result feeds fit() without another copy
    from flexynesis_amd.ingest import to_dataset
    from flexynesis_amd.models import DirectPred
    from flexynesis_amd.fit import fit
    y = (out[:, :16].sum(1) / 4).cpu()
    ds = to_dataset({"gex": out[:, :4096].contiguous()}, {"y": y}, {"y": "numerical"},
                    {"gex": [f"g{i}" for i in range(4096)]}, [f"s{i}" for i in range(N)])
    assert ds.dat["gex"].is_cuda
    model = DirectPred({"latent_dim": 16, "hidden_dim_factor": 0.1, "supervisor_hidden_dim": 8, "lr": 1e-3, "batch_size": 128,
                        "epochs": 2}, ds, ["y"], device_type="cuda")
    r = fit(model, ds, np.arange(0, 1792), np.arange(1792, N), batch_size=128, epochs=2, lr=1e-3, patience=5, seed=0)
    assert np.isfinite(r.val_loss) and r.steps == 2 * (1792 // 128)


def test_strided_device_views_are_ingested_in_place():
    """A column window of a wider resident matrix (leading dimension > n_features, base address not 16-byte aligned)
    goes through the scalar-load path without a copy and gives the same result as the compact matrix."""
    rng = np.random.default_rng(31)
    N, F = 120, 333
    big = rng.normal(size=(N, F + 10)) * 2 + 1
    big[rng.integers(0, N, 60), rng.integers(0, F + 10, 60)] = np.nan
    for dt in (torch.float32, torch.float64):
        wide = torch.from_numpy(big).to(dt).cuda()
        view = wide[:, 3:3 + F]
        assert view.stride(0) == F + 10 and view.data_ptr() % 16 != 0
        a = _imp().import_matrices({"x": view})
        b = _imp().import_matrices({"x": view.contiguous()})
        assert np.array_equal(a.features["x"], b.features["x"]) and np.array_equal(a.train_rows, b.train_rows)
        assert torch.equal(a.train["x"], b.train["x"])
        out = R.import_matrices({"x": big[:, 3:3 + F].astype(np.float32 if dt == torch.float32 else np.float64)})
        close_fp32(a.train["x"], out["train"]["x"], what=str(dt), min_exact=0.999)
