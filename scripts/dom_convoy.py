"""Are the workgroups of the dominant kernel in lockstep chip-wide?  (-DFT_PROFILE build.)  Every workgroup stamps the wall clock at
the start of its tiles 2, 5, 8, ...; if phases were spread evenly, the stamps of one tile index modulo the tile period would be
uniform; a chip-wide convoy shows as a cluster.   FXHIP_LIB=build_tmp/libfxhip_prof.so FX_FUSED_V1=1 python scripts/dom_convoy.py"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops, _lib

dev = torch.device("cuda:0")
n_out, k_in, B = 5000, 20000, 128
mapping = int(sys.argv[1]) if len(sys.argv) > 1 else 1
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
fn = _lib.lib.fx_debug_dw_adam_fwd_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (1024 * 8))()
for _ in range(3):
    ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs, mapping=mapping)
torch.cuda.synchronize()
fn(buf)
nwg = 512 if not int(__import__("os").environ.get("FX_FUSED_RUNS", "0")) else 79 * int(__import__("os").environ["FX_FUSED_RUNS"])
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:nwg].astype(np.float64) * 1e-2     # us
K = 6                                           # stamps 0..K-1 = tiles 2, 5, ..., 17 exist for every workgroup (>= 19 tiles each)
st = st[:, :K]
period = np.diff(st, axis=1).mean() / 3.0
print(f"mapping {mapping}: {nwg} workgroups, mean tile period {period:.2f} us")
t0 = st[:, 0].min()
q = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 5, 25, 50, 75, 95, 100)]
for k in range(K):
    print(f"  start of tile {3 * k + 2:2d} (us after the first workgroup's tile 2), percentiles 0/5/25/50/75/95/100: {q(st[:, k] - t0)}")
per = (st[:, K - 1] - st[:, 0]) / (3.0 * (K - 1))
print("  per-workgroup tile period (us):", q(per))
half = nwg // 2
print(f"  first half of the grid (arrives first on its CU): period {per[:half].mean():.2f}; second half: {per[half:].mean():.2f}")
# column-tile spread inside an XCD at one instant: how many iterations apart are its workgroups when the median one starts tile 11?
tmid = np.median(st[:, 3])
it_at = np.array([np.interp(tmid, st[i], 3.0 * np.arange(K) + 2) for i in range(nwg)])
for x in range(8):
    v = it_at[x::8]
    print(f"  XCD {x}: iteration reached when the median workgroup starts tile 11: min {v.min():5.1f} max {v.max():5.1f} std {v.std():4.2f}")
