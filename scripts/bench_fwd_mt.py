import os, sys, torch
sys.path.insert(0, "/root/repo")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
M, K, N = 384, 30000, 7500
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.01; W2 = W.clone(); b = torch.zeros(N, device=dev)
xs = ops.new_split_kb(M, K, dev); ops.split_bf16(ops.IMMEDIATE, xs[0], xs[1], X)
Y = torch.empty(M, N, device=dev); ws = ops.Workspace(dev)
def timeit(fn, n=10):
    fn(0); fn(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t = timeit(lambda i: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, xs[0], xs[1], W if i % 2 == 0 else W2, b, ws))
print(f"MT={os.environ.get('FX_FWD_MT','1')} splitk={os.environ.get('FX_SPLITK','auto')}: fwd+reduce {t:7.1f} us  ({N*K*4/t/1e6:.2f} TB/s of W)  splitk={ops.lib.fx_linear_fwd_bf16x3_splitk(M,N,K)}")
