"""Mapping x priority variants of the dominant kernel IN ONE PROCESS, on one placement of W / m / v chosen by the ParamStore search
(process-to-process comparisons carry the placement lottery: DESIGN.md section 3.10).   python scripts/dom_variants.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.engine import ParamStore

dev = torch.device("cuda:0")
n_out, k_in, B = 5000, 20000, 128
spec = ArchSpec("DirectPred", [("gex", k_in)], 64, 0.25, 16, [("y", "numerical", 1)], None, None, True)
st = ParamStore(spec, dev, materialize_big_grads=False)
key = "encoders.0.layer_1.weight"
print("placement:", st.placement.get(key))
W, m, v = st.p(key), st.m(key), st.v(key)
g = torch.Generator(device=dev); g.manual_seed(1)
m.copy_(torch.randn(n_out, k_in, generator=g, device=dev) * 1e-3); v.copy_(torch.rand(n_out, k_in, generator=g, device=dev) * 1e-5)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
S = max(ops.dw_adam_fwd_slabs(n_out, k_in, 128, mp) for mp in (1, 2, 3))
slabs = torch.zeros(S, B, n_out, device=dev)
dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2; x = torch.randn(B, k_in, generator=g, device=dev)
dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, x)


def time_it(mapping, prio, iters=30):
    ops.TUNE["fused_prio"] = prio
    def launch():
        ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs, mapping=mapping)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rep in range(3):
    row = []
    for mapping in (1, 2, 3):
        for prio in (0, 1, 2, 3):
            row.append(f"m{mapping}p{prio} {time_it(mapping, prio):6.1f}")
    print(" | ".join(row))
