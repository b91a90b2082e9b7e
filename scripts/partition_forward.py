"""W-only kernels (the unfused wide forward at M = 128 / 64 / 32, the data-gradient product) with W inside ONE partition against W lying ACROSS a
partition boundary of one big allocation (boundary located by pair probes).   python scripts/partition_forward.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
G = 140
N, K = 10000, 20000
words = N * K
buf = torch.zeros((G << 30) // 4, dtype=torch.float32, device=dev)
GBb = 1 << 30
def view(off_gb, n=N, k=K):
    o = int(off_gb * GBb) // 4 // 64 * 64
    return buf[o:o + n * k].view(n, k)
def pair(x, y):
    return 16.0 * 5000 * 20000 / ops.placement_probe_us(view(x, 5000, 20000), view(y, 5000, 20000), None) / 1e6
lo, hi = 0.0, None
for y in range(4, G - 2, 4):
    if pair(0, y) >= 5.6:
        hi = float(y); break
    lo = float(y)
if hi is None:
    print("no boundary in", G, "GB"); sys.exit(0)
while hi - lo > 0.05:
    mid = (lo + hi) / 2
    if pair(0, mid) >= 5.6: hi = mid
    else: lo = mid
print(f"boundary at ~{hi:.2f} GB of this allocation", flush=True)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
arr_gb = words * 4 / GBb
spots = [("inside one partition", 1.0), ("across the boundary (half / half)", hi - arr_gb / 2), ("across (a quarter / three quarters)", hi - arr_gb / 4)]
for M in (128, 64, 32):
    X = torch.randn(M, K, device=dev); b = torch.zeros(N, device=dev); Y = torch.empty(M, N, device=dev); ws = ops.Workspace(dev)
    s = ops.new_split_kb(M, K, dev); ops.split_bf16(ops.IMMEDIATE, s[0], s[1], X)
    dY = torch.randn(M, N, device=dev); dX = torch.empty(M, K, device=dev)
    row = []
    for rep in range(2):
        for name, off in spots:
            W = view(off); W.normal_(0, 0.01)
            t = timeit(lambda: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, s[0], s[1], W, b, ws))
            t2 = timeit(lambda: ops.linear_bwd_x(ops.IMMEDIATE, dX, dY, W, ws)) if M == 128 and hasattr(ops, "linear_bwd_x") else float("nan")
            print(f"M={M:3d} W {name:36s}: forward {t:6.1f} us ({N * K * 4 / t / 1e6:4.2f} TB/s)   data gradient {t2:6.1f} us", flush=True)
