"""fx_mmd_rows: first kernel vs the LDS-tiled one, stand-alone (us per launch).   python scripts/bench_mmd.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
for (P, B, L) in ((200, 128, 64), (200, 128, 128), (200, 32, 16)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    prior, z = torch.randn(P, L, generator=g, device=dev), torch.randn(B, L, generator=g, device=dev)
    rs, dz = torch.zeros(2 * (P + B), device=dev), torch.zeros(B, L, device=dev)
    for tiled in (False, True):
        for _ in range(5):
            ops.mmd_rows(ops.IMMEDIATE, rs, dz, prior, z, tiled=tiled)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.mmd_rows(ops.IMMEDIATE, rs, dz, prior, z, tiled=tiled)
        e1.record(); torch.cuda.synchronize()
        print(f"P={P} B={B} L={L} tiled={tiled}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us")
