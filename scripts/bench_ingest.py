"""Device ingest at cfg2 layer size (2048 x 20000): per-kernel time and effective HBM rate, the end-to-end
import_matrices time with the matrix already in HBM and from host memory, and the CPU restatement beside it.

    python scripts/bench_ingest.py [--n 2048] [--f 20000] [--dtype f32|f64] [--cpu]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd import ops  # noqa: E402
from flexynesis_amd.ingest import DeviceImporter  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--f", type=int, default=20000)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU restatement (oracle) on the host cores")
    a = ap.parse_args()
    N, F = a.n, a.f
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    es = 4 if a.dtype == "f32" else 8
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn(N, F, generator=g, device="cuda") * 2 + 5).to(dt)
    x[torch.randint(0, N, (F // 4,), generator=g, device="cuda"), torch.randint(0, F, (F // 4,), generator=g, device="cuda")] = float("nan")
    rec = ops.ImmediateRecorder()
    cols = torch.arange(F, dtype=torch.int32, device="cuda")
    rows = torch.arange(N, dtype=torch.int32, device="cuda")
    med = torch.full((F,), float("nan"), dtype=torch.float64, device="cuda")
    nan_cols = torch.nonzero(torch.isnan(x).any(0)).flatten().to(torch.int32)
    out = torch.empty((N, F), dtype=torch.float32, device="cuda")
    mean = torch.zeros(F, dtype=torch.float64, device="cuda")
    scale = torch.ones(F, dtype=torch.float64, device="cuda")
    mat = N * F * es
    t = timed(lambda: ops.col_moments(rec, x))
    print(f"fx_col_moments       {t*1e6:8.1f} us  {mat/t/1e9:7.1f} GB/s  (reads {mat/1e6:.0f} MB)")
    t = timed(lambda: ops.col_moments(rec, x, rows=rows, med=med, log1p=True))
    print(f"fx_col_moments+log1p {t*1e6:8.1f} us  {mat/t/1e9:7.1f} GB/s")
    t = timed(lambda: ops.col_median(rec, x, nan_cols, med))
    print(f"fx_col_median        {t*1e6:8.1f} us  ({nan_cols.numel()} columns with NaN)")
    t = timed(lambda: ops.row_moments(rec, x, cols, med))
    print(f"fx_row_moments       {t*1e6:8.1f} us  {mat/t/1e9:7.1f} GB/s")
    t = timed(lambda: ops.ingest_transform(rec, x, out, rows=rows, cols=cols, med=med, mean=mean, scale=scale))
    print(f"fx_ingest_transform  {t*1e6:8.1f} us  {(mat+N*F*4)/t/1e9:7.1f} GB/s  (reads {mat/1e6:.0f} MB, writes {N*F*4/1e6:.0f} MB)")
    imp = DeviceImporter()
    imp.import_matrices({"gex": x})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        imp.import_matrices({"gex": x})
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 5
    print(f"import_matrices, matrix resident in HBM: {t_dev*1e3:.2f} ms  ({N/t_dev:,.0f} samples/s)")
    xh = x.cpu().numpy()
    imp.import_matrices({"gex": xh})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        imp.import_matrices({"gex": xh})
    torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) / 3
    print(f"import_matrices, matrix in host memory (upload over PCIe included): {t_host*1e3:.2f} ms  ({mat/t_host/1e9:.1f} GB/s end to end)")
    try:
        import tempfile
        from flexynesis_amd import h5io
        h5io.lib()
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "gex.h5")
            h5io.write_modality_h5(path, xh.astype(np.float32), [f"s{i}" for i in range(N)], [f"f{i}" for i in range(F)])
            h5io.read_matrix_to_device(path)
            t0 = time.perf_counter()
            for _ in range(3):
                d, _, _ = h5io.read_matrix_to_device(path)
            torch.cuda.synchronize()
            t_h5 = (time.perf_counter() - t0) / 3
            t0 = time.perf_counter()
            for _ in range(3):
                h5io.read_modality_h5(path)
            t_h5h = (time.perf_counter() - t0) / 3
            print(f".h5 (page cache) -> pinned blocks -> HBM: {t_h5*1e3:.1f} ms ({N*F*4/t_h5/1e9:.2f} GB/s); the same file into pageable host memory only: {t_h5h*1e3:.1f} ms")
    except Exception as e:  # noqa: BLE001
        print("h5 leg skipped:", e)
    if a.cpu:
        from oracle import ingest_restate as R
        t0 = time.perf_counter()
        R.import_matrices({"gex": xh})
        t_cpu = time.perf_counter() - t0
        print(f"CPU restatement (numpy, {os.cpu_count()} cores visible): {t_cpu*1e3:.0f} ms  -> device path {t_cpu/t_dev:.0f}x (resident), {t_cpu/t_host:.0f}x (from host)")


if __name__ == "__main__":
    main()
