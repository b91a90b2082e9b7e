"""Stacked-rows forward (batches of more than 128 rows): the round-3 kernel with X by LDS-DMA (FX_FWD_MT=3) against the kernel that holds its
X fragments in registers (the default).  python scripts/fwd_reg_ab.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = ((384, 7500, 30000), (384, 7500, 30000), (256, 7500, 30000), (200, 5001, 10003), (500, 3000, 20000), (1024, 5000, 20000), (384, 40000, 5000), (128, 20000, 5000), (128, 5000, 20000), (128, 10000, 20000), (100, 20001, 5003), (64, 10000, 20000), (32, 10000, 20000), (32, 5000, 20000))
import os
for it, (M, N, K) in enumerate(shapes[:int(os.environ.get("FX_AB_ROWS", len(shapes)))]):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.01; b = torch.randn(N, device=dev)
    if it == 1: X.zero_(); W.zero_()
    Y = torch.empty(M, N, device=dev); ws = ops.Workspace(dev)
    s = ops.new_split_kb(M, K, dev); ops.split_bf16(ops.IMMEDIATE, s[0], s[1], X)
    out = {}
    for mode in ((3, 0) if M > 128 else (0, 4)):
        ops.TUNE["fwd_no_mt"] = mode
        Y.fill_(float("nan"))
        t = timeit(lambda: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, s[0], s[1], W, b, ws))
        out[mode] = (t, Y.clone())
    ref = (X.double() @ W.double().t() + b.double()).float()
    (ta, ya), (tb, yb) = out.values()
    print(f"M={M:4d} N={N:6d} K={K:6d}{' zeros' if it == 1 else '      '}  X by LDS-DMA {ta:7.1f} us   X in registers {tb:7.1f} us ({2 * 3 * M * N * K / tb / 1e9:5.2f} PFLOP/s-eq {N * K * 4 / tb / 1e6:5.2f} TB/s)"
          f"   max err {(ya - ref).abs().max().item():.2e} / {(yb - ref).abs().max().item():.2e}")
