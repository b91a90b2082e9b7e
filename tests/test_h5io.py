"""HDF5 modality files (reference csv_to_h5.py:13-21 layout) over the HDF5 C library: write -> the library's own
h5dump -> read back, on the CPU; streaming into HBM and on through the device ingest on the GPU."""
import os
import shutil
import subprocess

import numpy as np
import pytest

h5io = pytest.importorskip("flexynesis_amd.h5io")

try:
    h5io.lib()
    HAVE = True
except h5io.H5Error:
    HAVE = False
needs_hdf5 = pytest.mark.skipif(not HAVE, reason="HDF5 C library not present")


def make(tmp_path, n=37, F=53, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, F)).astype(np.float32)
    X[3, 5] = np.nan
    samples = [f"TCGA-{i:04d}" for i in range(n)]
    feats = [f"ENSG{i:011d}" if i % 3 else f"g{i}" for i in range(F)]          # mixed lengths: NUL padding is stripped
    p = str(tmp_path / "gex.h5")
    h5io.write_modality_h5(p, X, samples, feats)
    return p, X, samples, feats


@needs_hdf5
def test_roundtrip_and_layout(tmp_path):
    p, X, samples, feats = make(tmp_path)
    M, s, f = h5io.read_modality_h5(p)
    assert M.dtype == np.float32 and np.array_equal(M, X, equal_nan=True)
    assert s == samples and f == feats
    with h5io.H5File(p) as h:
        shape, dt, chunk = h.matrix_info()
        assert shape == X.shape and dt == np.float32 and chunk == (1, X.shape[1])      # csv_to_h5.py:107
        part = np.empty((5, X.shape[1]), dtype=np.float32)
        h.read_rows_into(part.ctypes.data, 11, 16, X.shape[1], np.float32)
        assert np.array_equal(part, X[11:16], equal_nan=True)


@needs_hdf5
def test_variable_length_strings(tmp_path):
    """h5py stores ``str`` data as variable-length strings (char pointers on read), unlike the converter's 'S' dtype."""
    p = str(tmp_path / "vlen.h5")
    samples = ["a", "TCGA-0001-long-name", "", "s3"]
    feats = [f"gene{i}" * (1 + i % 3) for i in range(7)]
    with h5io.H5File(p, "w") as f:
        f.write_matrix(np.zeros((4, 7), dtype=np.float32), "matrix", chunks=(1, 7))
        f.write_strings("sample_ids", samples, variable=True)
        f.write_strings("feature_names", feats, variable=True)
    M, s, ft = h5io.read_modality_h5(p)
    assert s == samples and ft == feats and M.shape == (4, 7)


@needs_hdf5
def test_file_is_what_the_hdf5_tools_see(tmp_path):
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        pytest.skip("h5dump not installed")
    p, X, samples, feats = make(tmp_path, n=4, F=6)
    hdr = subprocess.run([h5dump, "-H", "-p", p], capture_output=True, text=True, check=True).stdout
    assert 'DATASET "matrix"' in hdr and "H5T_IEEE_F32LE" in hdr and "( 4, 6 ) / ( 4, 6 )" in hdr
    assert "CHUNKED ( 1, 6 )" in hdr
    assert 'DATASET "sample_ids"' in hdr and 'DATASET "feature_names"' in hdr and "H5T_STRING" in hdr
    body = subprocess.run([h5dump, "-d", "/sample_ids", p], capture_output=True, text=True, check=True).stdout
    assert all(s in body for s in samples)


@needs_hdf5
def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        h5io.read_modality_h5(str(tmp_path / "missing.h5"))
    bad = tmp_path / "not_hdf5.h5"
    bad.write_bytes(b"feature,s1,s2\n")
    with pytest.raises(h5io.H5Error):
        h5io.read_modality_h5(str(bad))
    with pytest.raises(h5io.H5Error):
        h5io.write_modality_h5(str(tmp_path / "x.h5"), np.zeros((3, 4), np.float32), ["a", "b"], ["f"] * 4)
    p = str(tmp_path / "nomatrix.h5")
    with h5io.H5File(p, "w") as h:
        h.write_strings("sample_ids", ["a"])
    with pytest.raises(h5io.H5Error):
        h5io.read_modality_h5(p)


@pytest.mark.gpu
def test_stream_to_hbm_and_ingest(tmp_path):
    """(Not skipped when the HDF5 library is missing: on the GPU box the .h5 path must work or fail loudly.)"""
    assert HAVE, "the HDF5 C library is missing on the GPU box: .h5 modality files cannot be streamed into HBM"
    import torch
    from flexynesis_amd.ingest import DeviceImporter
    from oracle import ingest_restate as R
    rng = np.random.default_rng(4)
    n, F = 301, 2000
    X = (rng.normal(size=(n, F)) * rng.uniform(0.1, 3, size=(1, F)) + 4).astype(np.float32)
    X[rng.integers(0, n, 300), rng.integers(0, F, 300)] = np.nan
    samples, feats = [f"s{i}" for i in range(n)], [f"f{i}" for i in range(F)]
    p = str(tmp_path / "gex.h5")
    h5io.write_modality_h5(p, X, samples, feats)
    for block in (1 << 30, 40 * F * 4, 7 * F * 4):                    # one block, 8 blocks, 43 blocks
        d, s, f = h5io.read_matrix_to_device(p, "cuda:0", block_bytes=block)
        assert d.is_cuda and d.dtype == torch.float32 and (s, f) == (samples, feats)
        assert np.array_equal(d.cpu().numpy(), X, equal_nan=True)
    res = DeviceImporter().import_matrices({"gex": d})
    out = R.import_matrices({"gex": X})
    assert np.array_equal(res.features["gex"], out["features"]["gex"]) and np.array_equal(res.train_rows, out["train_rows"])
    got, exp = res.train["gex"].cpu().numpy(), out["train"]["gex"]
    assert np.abs(got - exp).max() < 1e-6 and (got == exp).mean() > 0.999
