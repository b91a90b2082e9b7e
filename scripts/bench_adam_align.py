"""Does the relative placement of W / m / v in memory change the dW+Adam kernel's speed?  (In the step, the first
wide weight's launch takes 480 us and the second's 411 us, every step.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd import ops
dev = torch.device("cuda:0")
B, F, H = 128, 20000, 5000
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn(B, F, device=dev, generator=g)
dY = torch.randn(B, H, device=dev, generator=g) * 1e-3
xt = ops.new_split(F, B, dev); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], X)
dyt = ops.new_split(H, B, dev); ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dY)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
n = H * F
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
flush = torch.empty(300 * 1024 * 1024 // 4, device=dev)
for pad_bytes in (0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, 2 << 20, (2 << 20) + 8192):
    pad = pad_bytes // 4
    flat = torch.zeros(3 * n + 2 * pad + 1024, device=dev)
    base = (-flat.data_ptr() // 4) % 64          # 256-B align the first tensor
    W = flat[base:base + n].view(H, F); M = flat[base + n + pad:base + 2 * n + pad].view(H, F)
    V = flat[base + 2 * n + 2 * pad:base + 3 * n + 2 * pad].view(H, F)
    W.normal_(0, 0.01)
    t = timeit(lambda: ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W, M, V, dyt[0], dyt[1], xt[0], xt[1], ctrl))
    print(f"pad {pad_bytes:>9d} B   W%2M={W.data_ptr() % (2 << 20):>8d} spacing {(M.data_ptr() - W.data_ptr())} : {t:7.1f} us = {24 * n / t / 1e6:.2f} TB/s", flush=True)
    del flat, W, M, V
# separately allocated (what ParamStore does)
for trial in range(3):
    W = torch.randn(H, F, device=dev) * 0.01; M = torch.zeros(H, F, device=dev); V = torch.zeros(H, F, device=dev)
    t = timeit(lambda: ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W, M, V, dyt[0], dyt[1], xt[0], xt[1], ctrl))
    print(f"separate allocs: ptr%2M W={W.data_ptr() % (2 << 20)} M-W={M.data_ptr() - W.data_ptr()} V-M={V.data_ptr() - M.data_ptr()} : {t:7.1f} us", flush=True)
    keep = (W, M, V) if trial == 0 else None
