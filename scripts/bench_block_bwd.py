"""Duration of fx_block_bwd at cfg2 shapes (B=128, H=5000, L=64), HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd import ops
dev = torch.device("cuda:0")
B, C, L = 128, 5000, 64
x = torch.randn(B, C, device=dev); out = torch.relu(torch.randn(B, C, device=dev))
dE = torch.randn(B, L, device=dev); W = torch.randn(L, C, device=dev); gW = torch.empty(L, C, device=dev); gb = torch.empty(L, device=dev)
gamma = torch.ones(C, device=dev); sm = torch.zeros(C, device=dev); si = torch.ones(C, device=dev)
dg, db, dbias = (torch.empty(C, device=dev) for _ in range(3))
dyT = ops.new_split(C, B, dev); gx = torch.randn(B, B, device=dev)
slots = torch.zeros(ops.block_bwd_blocks(C), dtype=torch.float64, device=dev)
def run():
    ops.block_bwd(ops.IMMEDIATE, [(dE, W, gW, gb)], x, out, gamma, sm, si, dg, db, dbias, 0, 2, 0.1, dyT=dyT, gram_x=gx, slots=slots)
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(round(1e3 * e0.elapsed_time(e1), 1))
print("fx_block_bwd us:", ts)
