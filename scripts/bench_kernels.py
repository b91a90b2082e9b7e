"""Standalone timing of the two wide-layer kernels at cfg2 shapes (HIP events, N launches each)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd import ops
dev = torch.device("cuda:0")
B, F, H = 128, 20000, 5000
n = int(os.environ.get("N", "20"))
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn(B, F, device=dev, generator=g)
W = torch.randn(H, F, device=dev, generator=g) * 0.01
M = torch.zeros_like(W); V = torch.zeros_like(W)
b = torch.zeros(H, device=dev)
dY = torch.randn(B, H, device=dev, generator=g) * 1e-3
Y = torch.empty(B, H, device=dev)
ws = ops.Workspace(dev)
xs = ops.new_split_kb(B, F, dev); ops.split_bf16(ops.IMMEDIATE, xs[0], xs[1], X)
xt = ops.new_split(F, B, dev); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], X)
dyt = ops.new_split(H, B, dev); ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dY)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
# second weight so consecutive launches do not hit a warm MALL
W2 = W.clone(); M2 = torch.zeros_like(W); V2 = torch.zeros_like(W)
def timeit(fn, n):
    fn(0); fn(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fwd = lambda i: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, xs[0], xs[1], W if i % 2 == 0 else W2, b, ws)
adam = lambda i: ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, *( (W, M, V) if i % 2 == 0 else (W2, M2, V2)), dyt[0], dyt[1], xt[0], xt[1], ctrl)
tf, ta = timeit(fwd, n), timeit(adam, n)
print(f"env NT_FWD={os.environ.get('FX_NT_FWD','0')} NT_ADAM={os.environ.get('FX_NT_ADAM','0')} SPLITK={os.environ.get('FX_SPLITK','auto')}: "
      f"fwd(+reduce) {tf:.1f} us = {H*F*4/tf/1e6:.2f} TB/s ; dw+adam {ta:.1f} us = {24*H*F/ta/1e6:.2f} TB/s", flush=True)
G = torch.randn(H, F, device=dev, generator=g) * 1e-3
flat = lambda i: ops.adam_flat(ops.IMMEDIATE, (W if i % 2 == 0 else W2).view(-1), G.view(-1), (M if i % 2 == 0 else M2).view(-1), (V if i % 2 == 0 else V2).view(-1), ctrl)
tfl = timeit(flat, n)
print(f"adam_flat (28 B/param, dword accesses): {tfl:.1f} us = {28*H*F/tfl/1e6:.2f} TB/s", flush=True)
cp = lambda i: (W2 if i % 2 == 0 else W).copy_(M if i % 2 == 0 else M2)
tc = timeit(cp, n)
print(f"torch copy_ 400 MB (8 B/elem): {tc:.1f} us = {8*H*F/tc/1e6:.2f} TB/s", flush=True)
