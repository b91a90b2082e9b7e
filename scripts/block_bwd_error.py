"""Arithmetic error of fx_block_bwd against an fp64 evaluation on the same fp32-rounded inputs (dy, gW, the Gram norm share):
the fp32-MFMA products (round 3) and the FMA loops they replaced (FXHIP_LIB=<older build>) both measure 1.6e-7 .. 2.2e-7 rms, no bias."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from flexynesis_amd import ops
dev = torch.device("cuda:0")
for (B, C, Ls, pre, post) in [(128, 5000, (64,), 0, 2), (128, 1250, (100, 33), 1, 0)]:
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, C, generator=g, dtype=torch.float64)
    gamma = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64) * 0.1
    mask = (torch.rand(B, C, generator=g) < 0.9).double()
    Ws = [torch.randn(L, C, generator=g, dtype=torch.float64) / C ** 0.5 for L in Ls]
    dEs = [torch.randn(B, L, generator=g, dtype=torch.float64) for L in Ls]
    X = torch.randn(B, 300, generator=g, dtype=torch.float64)
    f32 = lambda t: t.float().to(dev).contiguous()
    # fp64 reference evaluated on the fp32-ROUNDED inputs (so only the kernel's arithmetic is measured)
    x, gamma, Ws, dEs = x.float().double(), gamma.float().double(), [w.float().double() for w in Ws], [d.float().double() for d in dEs]
    xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    Wr = [w.clone().requires_grad_(True) for w in Ws]
    h = torch.where(xr > 0, xr, 0.2 * xr) if pre == 1 else xr
    mean, var = h.mean(0), h.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    mean, invstd = mean.float().double(), invstd.float().double()
    yb = (h - mean) * invstd * gr + br
    out = torch.relu(yb) * (mask / 0.9) if post == 2 else yb
    outf = out.detach().float().double()
    # autograd through the rounded saved tensors is awkward; compute the reference by hand in fp64
    da = sum(d @ w for d, w in zip(dEs, Ws))
    if post == 2:
        da = torch.where(outf > 0, da / 0.9, torch.zeros_like(da))
    xa = torch.where(x > 0, x, 0.2 * x) if pre == 1 else x
    xh = (xa - mean) * invstd
    dy = gamma * invstd * (da - da.mean(0) - xh * (da * xh).mean(0))
    if pre == 1:
        dy = torch.where(x > 0, dy, 0.2 * dy)
    gWs = [d.t() @ outf for d in dEs]
    norm2 = float(((dy.t() @ X) ** 2).sum())
    ups = [(f32(d), f32(w), torch.empty(w.shape, device=dev), torch.empty((w.shape[0],), device=dev)) for w, d in zip(Ws, dEs)]
    dg, db, dbias = (torch.empty(C, device=dev) for _ in range(3))
    dyo = torch.empty(B, C, device=dev)
    dyT = ops.new_split(C, B, dev)
    gx = f32(X @ X.t())
    slots = torch.zeros(ops.block_bwd_blocks(C), dtype=torch.float64, device=dev)
    ops.block_bwd(ops.IMMEDIATE, ups, f32(x), f32(outf), f32(gamma), f32(mean), f32(invstd), dg, db, dbias, pre, post,
                  0.1 if post == 2 else 0.0, dy=dyo, dyT=dyT, gram_x=gx, slots=slots)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    mx = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    print(os.environ.get("FXHIP_LIB", "default")[-14:], (B, C, Ls), "dy rel-rms %.2e max %.2e | gW rel-rms %s | norm2 rel %.2e | mean signed dy err %.2e" % (
        rel(dyo, dy), mx(dyo, dy), ["%.2e" % rel(u[2], gw) for u, gw in zip(ups, gWs)], abs(float(slots.sum()) - norm2) / norm2,
        float((dyo.double().cpu() - dy).mean() / dy.abs().mean())))
