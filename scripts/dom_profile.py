"""Per-phase time of the dominant kernel's tile loop, from the -DFT_PROFILE build (scripts/build_variant.py prof ...):
FXHIP_LIB=build_tmp/libfxhip_prof.so python scripts/dom_profile.py [mapping]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops, _lib

dev = torch.device("cuda:0")
n_out, k_in, B = 5000, 20000, 128
mapping = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device=dev); g.manual_seed(1)
ldw = ops.pad32(k_in)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
S = max(ops.dw_adam_fwd_slabs(n_out, k_in, 128, m_) for m_ in (1, 2, 3))
slabs = torch.zeros(S, B, n_out, device=dev)
dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2
x = torch.randn(B, k_in, generator=g, device=dev); xn = torch.randn(B, k_in, generator=g, device=dev)
dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
W = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
m = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-3
v = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-5
fn = _lib.lib.fx_debug_dw_adam_fwd_profile
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()


def launch():
    ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs,
                                  mapping=mapping, nt=bool(nt))


for _ in range(3):
    launch()
torch.cuda.synchronize()
fn(buf, 1)
it = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it):
    launch()
e1.record()
torch.cuda.synchronize()
fn(buf, 1)
us = e0.elapsed_time(e1) / it * 1e3
tiles = ((n_out + 63) // 64) * ((k_in + 127) // 128)
names = ["tile start -> first DMA issued", "K-loop: barrier waits (4 / tile)", "K-loop: ds_read + MFMA (4 / tile)", "K-loop: closing barrier",
         "transpose store + barrier", "phase 3: issue 12 loads", "phase 3: wait for W / m / v", "phase 3: Adam + stores + W_new -> LDS", "phase 3: closing barrier",
         "forward: barrier waits (4 / tile)", "forward: ds_read + MFMA (4 / tile)", "forward: closing barrier", "slab store (per run)",
         "K-loop: own vmcnt(0) before the barrier (4 / tile)", "phase 3: own vmcnt(0) = store drain", "forward: own vmcnt(0) before the barrier (4 / tile)"]
print(f"mapping {mapping} nt={nt}: {us:.1f} us per launch (instrumented); wave 0 of each workgroup, us per TILE (x {tiles / (((n_out + 63) // 64) * S):.1f} tiles per workgroup)")
tot = 0.0
for i, nme in enumerate(names):
    per_tile = buf[i] / it / tiles * 1e-2        # 100 MHz ticks -> us
    tot += per_tile
    print(f"  {nme:42s} {per_tile:7.3f}")
print(f"  {'sum':42s} {tot:7.3f}   (x tiles per workgroup = {tot * tiles / (((n_out + 63) // 64) * S):.1f} us)")
