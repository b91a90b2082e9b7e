"""Pin the ingest restatement (oracle/ingest_restate.py) against the fixtures generated from the reference's own
DataImporter methods (tests/golden/ingest_*.npz, oracle/gen_ingest_goldens.py), against pandas / scikit-learn (the
un-vendored dependencies whose arithmetic it restates) and -- when /root/reference is present -- against the reference
itself on fresh seeds.  CPU only."""
import os

import numpy as np
import pytest

from oracle import ingest_restate as R
from oracle import ref_shim

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["ingest_f64", "ingest_f32", "ingest_f64_log", "ingest_f32_log"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    layers = [str(x) for x in z["layers"]]
    train = {k: z[f"in_train/{k}"] for k in layers}
    test = {k: z[f"in_test/{k}"] for k in layers}
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp/")}
    return layers, train, test, exp, float(z["thresholds"][0]), float(z["thresholds"][1]), bool(z["log_transform"])


def check_against(out, exp, layers, exact=True):
    assert np.array_equal(out["train_rows"], exp["train_rows"])
    assert np.array_equal(out["test_rows"], exp["test_rows"])
    for k in layers:
        assert np.array_equal(out["features"][k], exp[f"features/{k}"]), k
        for side in ("train", "test"):
            a, b = out[side][k], exp[f"{side}/{k}"]
            assert a.dtype == np.float32 and a.shape == b.shape, (k, side, a.shape, b.shape)
            if exact:
                assert np.array_equal(a, b), (k, side, float(np.abs(a - b).max()))
            else:
                np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out["scalers"][k][0], exp[f"mean/{k}"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(out["scalers"][k][1], exp[f"scale/{k}"], rtol=1e-13)


@pytest.mark.parametrize("case", CASES)
def test_restatement_reproduces_reference_fixture(case):
    layers, train, test, exp, vt, nat, logt = load_case(case)
    out = R.import_matrices(train, test, variance_threshold=vt, na_threshold=nat, log_transform=logt)
    check_against(out, exp, layers, exact=True)            # bit-exact on the fixtures
    assert exp["train_rows"].size < next(iter(train.values())).shape[0]      # the constant sample was dropped
    for k in layers:                                                         # filters and harmonize did something
        assert exp[f"features/{k}"].size < train[k].shape[1]


def test_fixture_covers_the_awkward_cases():
    layers, train, test, exp, *_ = load_case("ingest_f64")
    X = train["gex"]
    assert np.isnan(X).any() and (np.delete(X[:, 3], X.shape[0] // 2) == 1.0).all()
    kept = set(exp["features/gex"].tolist())
    assert 3 not in kept and 9 not in kept and 7 in kept and 11 not in kept      # low variance / too many NaN / imputed / test-only constant
    assert np.isfinite(exp["train/gex"]).all() and np.isfinite(exp["test/gex"]).all()


def test_pieces_match_pandas_and_sklearn():
    pd = pytest.importorskip("pandas")
    skp = pytest.importorskip("sklearn.preprocessing")
    rng = np.random.default_rng(5)
    for dtype in (np.float64, np.float32):
        X = (rng.normal(size=(53, 40)) * rng.uniform(0.1, 4, size=(1, 40)) + rng.normal(size=(1, 40)) * 50).astype(dtype)
        X[rng.integers(0, 53, 30), rng.integers(0, 40, 30)] = np.nan
        X[:, 4] = 7.0
        df = pd.DataFrame(X.T)                                       # reference orientation: features as rows
        v = R.nanvar_ddof1(X, axis=0)
        pv = df.var(axis=1).values
        assert v.dtype == pv.dtype
        np.testing.assert_allclose(v, pv, rtol=1e-12 if dtype == np.float64 else 2e-7)
        assert R.quantile_linear(pv, 0.01) == pytest.approx(df.var(axis=1).quantile(0.01), rel=1e-15)
        med = np.array([np.nanmedian(X[:, c]) for c in range(40)])
        assert np.array_equal(med, df.T.median(axis=0).values.astype(dtype), equal_nan=True)
        Xi = R.impute_median(X)
        sc = skp.StandardScaler().fit(Xi)
        mean, scale = R.scaler_fit(Xi)
        np.testing.assert_allclose(mean, sc.mean_, rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(scale, sc.scale_, rtol=1e-15)
        assert scale[4] == 1.0
        assert np.array_equal(R.scaler_transform(Xi, mean, scale), sc.transform(Xi).astype(np.float32))


def test_harmonize_keeps_training_order():
    a, b = R.harmonize(np.array([5, 2, 9, 7]), np.array([7, 5, 1, 9]))
    assert a.tolist() == [0, 2, 3] and b.tolist() == [1, 3, 0]


@pytest.mark.skipif(not ref_shim.available(), reason="reference not present (GPU box)")
@pytest.mark.parametrize("seed,dtype,logt", [(101, np.float64, False), (102, np.float32, False), (103, np.float64, True),
                                             (104, np.float32, True)])
def test_restatement_matches_live_reference(seed, dtype, logt):
    from oracle import gen_ingest_goldens as G
    feats = [("a", 60), ("b", 45)]
    train, test = G.make_case(seed, 33, 14, feats, dtype, counts=logt)
    exp = G.reference_import(train, test, 0.02, 0.2, logt)
    out = R.import_matrices(train, test, variance_threshold=0.02, na_threshold=0.2, log_transform=logt)
    check_against(out, exp, [k for k, _ in feats], exact=True)
