"""Where does the v2 fused kernel differ from the stand-alone dW + Adam kernel?  python scripts/debug_fused_v2.py [n_out k_in B]"""
import sys, torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
n_out, k_in, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5000, 20000, 128)
Bn = 128
g = torch.Generator(device=dev); g.manual_seed(n_out * 7 + k_in)
dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2
x = torch.randn(B, k_in, generator=g, device=dev)
xn = torch.randn(Bn, k_in, generator=g, device=dev)
ldw = ops.pad32(k_in)
W0 = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
m0 = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-3
v0 = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-5
ctrl = torch.zeros(64, device=dev); ctrl[0] = 6.0
ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3); ctrl[4] = 0.6
dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(Bn, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
W1, m1, v1 = W0.clone(), m0.clone(), v0.clone()
ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W1[:, :k_in], m1[:, :k_in], v1[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl)
S = max(ops.dw_adam_fwd_slabs(n_out, k_in, 128, m_) for m_ in (1, 2, 3))
for rep in range(3):
    for mapping in (1, 3):
        W2, m2, v2 = W0.clone(), m0.clone(), v0.clone()
        slabs = torch.full((S, Bn, n_out), float("nan"), device=dev)
        ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W2[:, :k_in], m2[:, :k_in], v2[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, Bn, slabs,
                                      mapping=mapping)
        torch.cuda.synchronize()
        for nm, a, b in (("W", W2, W1), ("m", m2, m1), ("v", v2, v1)):
            bad = (a != b)
            nb = int(bad.sum())
            if nb == 0:
                print(f"rep {rep} map {mapping} {nm}: identical")
                continue
            idx = bad.nonzero()
            rows, cols = idx[:, 0], idx[:, 1]
            tiles = torch.unique(torch.stack([rows // 64, cols // 128], 1), dim=0)
            rel = float(((a - b).abs().max()) / b.abs().max())
            print(f"rep {rep} map {mapping} {nm}: {nb} elements differ (max |diff| / max |ref| = {rel:.3e}), {tiles.shape[0]} tiles of {((n_out+63)//64)*((k_in+127)//128)}; "
                  f"first tiles (tm, tn): {tiles[:12].tolist()}; rows in tile of first 8: {(rows[:8] % 64).tolist()}, cols in tile: {(cols[:8] % 128).tolist()}")
        y = slabs.sum(0).double()
        ref = xn.double() @ W2[:, :k_in].double().t()
        print(f"   forward rel err {float((y - ref).norm() / ref.norm()):.3e}, nan slabs {int(torch.isnan(slabs).sum())}")
