"""Do the HBM-bound dW + Adam launch of one wide weight and the MFMA-bound stacked-rows forward through ANOTHER weight overlap when they are
issued on two streams?  (cfg4 shapes: [7500, 30000], K = M = 384).   python scripts/overlap_exp.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
n_out, k_in, B = 7500, 30000, 384
g = torch.Generator(device=dev); g.manual_seed(1)
ldw = ops.pad32(k_in)
def arr(scale): return (torch.randn(n_out, ldw, generator=g, device=dev) * scale)[:, :k_in]
W1, m1, v1 = arr(k_in ** -0.5), arr(1e-3), arr(1e-3); v1.abs_()
W2 = arr(k_in ** -0.5)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2; x = torch.randn(B, k_in, generator=g, device=dev)
dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, x)
y = torch.empty(B, n_out, device=dev); ws = ops.Workspace(dev); bias = torch.zeros(n_out, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def adam(): ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W1, m1, v1, dyt[0], dyt[1], xt[0], xt[1], ctrl)
def fwd(): ops.linear_fwd_bf16x3(ops.IMMEDIATE, y, xnh, xnl, W2, bias, ws)
def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def both(order):
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    first, second = ((s1, adam), (s2, fwd)) if order == 0 else ((s2, fwd), (s1, adam))
    for st, fn in (first, second):
        with torch.cuda.stream(st): fn()
    main.wait_stream(s1); main.wait_stream(s2)
ta, tf = timed(adam), timed(fwd)
print(f"dW + Adam alone {ta:.1f} us, forward alone {tf:.1f} us, sum {ta + tf:.1f}")
for order in (0, 1):
    print(f"two streams, {'Adam' if order == 0 else 'forward'} issued first: {timed(lambda: both(order)):.1f} us per pair")
