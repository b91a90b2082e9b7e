import sys, torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
n_out, k_in, B, Bn = 64, 128, 32, 128
g = torch.Generator(device=dev); g.manual_seed(1)
dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2
x = torch.randn(B, k_in, generator=g, device=dev)
xn = torch.randn(Bn, k_in, generator=g, device=dev)
ldw = ops.pad32(k_in)
W = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
m = torch.zeros_like(W); v = torch.zeros_like(W)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(Bn, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
S = ops.dw_adam_fwd_slabs(n_out, k_in)
slabs = torch.full((S, Bn, n_out), float("nan"), device=dev)
ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, Bn, slabs)
torch.cuda.synchronize()
y = slabs.sum(0).double()
Wd = W[:, :k_in].double()
ref = xn.double() @ Wd.t()
print("S", S, "rel err per row (first 40):", [round(float((y[i] - ref[i]).norm() / ref[i].norm()), 4) for i in range(40)])
# which reference row does each output row match best?
match = [(int(((ref - y[i]).norm(dim=1)).argmin())) for i in range(Bn)]
print("best matching ref row:", match)
# try: does y equal ref computed with permuted K chunks?
for name, perm in (("id", None),):
    pass
# column structure
print("rel err per col (first 16):", [round(float((y[:, j] - ref[:, j]).norm() / ref[:, j].norm()), 4) for j in range(16)])
