"""Is a placement that is slow under one workgroup mapping of the fused dW + Adam + forward kernel slow under the others too?
For P distinct placements of W / m / v ([5000, 20000] by default; all arrays stay allocated) the real kernel is timed under mapping
1 (plain), 2 (row blocks interleaved over the XCDs), 3 (XCD-contiguous row blocks: the default where it applies) and with 2 .. 8 runs per row block.
python scripts/mapping_vs_placement.py [N K P]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K, P = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 20000, 12)
B = 128
dy = torch.randn(B, N, device=dev) * 1e-2; x = torch.randn(B, K, device=dev); xn = torch.randn(B, K, device=dev)
dyt, xt = ops.new_split(N, B, dev), ops.new_split(K, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, K, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
ctrl = torch.zeros(64, device=dev); ctrl[0] = 9.0
ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3); ctrl[4] = 0.5
slabs = torch.zeros(16, B, N, device=dev)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)
pitch = ops.pad32(K)
keep = []
variants = [("map1", 1, 0), ("map2", 2, 0), ("map3", 3, 0), ("map1/r4", 1, 4), ("map1/r5", 1, 5), ("map1/r8", 1, 8), ("map3/r4", 3, 4), ("map3/r8", 3, 8)]
print("placement  probe us | " + " ".join(f"{v[0]:>8s}" for v in variants))
for p in range(P):
    arrs = [torch.zeros(N, pitch, device=dev) for _ in range(3)]
    keep.append(arrs)
    W, m, v = (a[:, :K] for a in arrs)
    W.normal_(0, 0.01)
    row = []
    for name, mapping, runs in variants:
        ops.TUNE["fused_runs"] = runs
        try:
            row.append(timeit(lambda: ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs, mapping=mapping)))
        except Exception as e:
            row.append(float("nan"))
    ops.TUNE["fused_runs"] = 0
    print(f"{p:9d}  {ops.placement_probe_us(W, m, v):8.1f} | " + " ".join(f"{t:8.1f}" for t in row), flush=True)
