"""Dominant kernel (fx_linear_dw_adam_fwd_bf16x3) stand-alone at the cfg2 shape: block mapping x operand data.

    python scripts/dom_diag.py [n_out k_in]

* mapping 1 (plain), 2 (XCD-grouped, interleaved row blocks), 3 (XCD-contiguous row blocks): does the number of distinct rows
  (= address translations) an XCD touches matter on this box?
* operands random vs all-zero, W / m / v random vs zero: is the cost of the two GEMM phases TIME (latency that is not hidden) or
  POWER (switching activity lowering the clocks)?  All-zero operands keep the instruction stream and every address identical.
"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops

dev = torch.device("cuda:0")
n_out, k_in = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 20000)
B = Bn = 128
g = torch.Generator(device=dev)
g.manual_seed(1)
ldw = ops.pad32(k_in)
ctrl = torch.zeros(64, device=dev)
ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
S = max(ops.dw_adam_fwd_slabs(n_out, k_in, 128, m_) for m_ in (1, 2, 3))
slabs = torch.zeros(S, Bn, n_out, device=dev)


def operands(zero):
    dy = torch.zeros(B, n_out, device=dev) if zero else torch.randn(B, n_out, generator=g, device=dev) * 1e-2
    x = torch.zeros(B, k_in, device=dev) if zero else torch.randn(B, k_in, generator=g, device=dev)
    xn = torch.zeros(Bn, k_in, device=dev) if zero else torch.randn(Bn, k_in, generator=g, device=dev)
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy)
    ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    xnh, xnl = ops.new_split_kb(Bn, k_in, dev)
    ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
    return dyt, xt, xnh, xnl


def weights(zero):
    if zero:
        return [torch.zeros(n_out, ldw, device=dev) for _ in range(3)]
    W = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
    m = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-3
    v = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-5
    return [W, m, v]


def time_it(wmv, op, mapping, iters=20):
    W, m, v = wmv
    dyt, xt, xnh, xnl = op

    def launch():
        ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl,
                                      Bn, slabs, mapping=mapping)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


op_r, op_z = operands(False), operands(True)
w_r, w_z = weights(False), weights(True)
bytes_launch = 24.0 * n_out * k_in
print(f"shape [{n_out}, {k_in}] B={B}  slabs={S}  (us per launch, TB/s of 24 B/param)")
import os
cases = (("W random, operands random", w_r, op_r), ("W random, operands ZERO", w_r, op_z),
         ("W zero,   operands ZERO", w_z, op_z), ("W zero,   operands random", w_z, op_r))
if os.environ.get("DOM_QUICK"):
    cases = cases[:1]
for rep in range(3 if os.environ.get("DOM_QUICK") else 2):
    for name, wmv, op in cases:
        row = []
        for mapping in (1, 2, 3):
            us = time_it(wmv, op, mapping)
            row.append(f"map{mapping} {us:7.1f} us {bytes_launch / us / 1e6:5.2f}")
        print(f"{name:28s} | " + " | ".join(row))
