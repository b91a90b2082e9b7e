"""One process, one box: the probe's rating of W / m / v ([5000, 20000]) when the three arrays are (a) separate allocations, (b) carved out of one
allocation of 1.2 / 2 / 4 / 8 GB -- alternating, every allocation kept.   python scripts/alloc_size_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K = 5000, 20000
pitch = ops.pad32(K); words = N * pitch; A = (words * 4 + (1 << 21) - 1) >> 21 << 21
keep = []
def rate(W, m, v):
    return 24.0 * N * K / ops.placement_probe_us(W, m, v) / 1e6
def separate():
    arrs = [torch.zeros(N, pitch, device=dev) for _ in range(3)]
    keep.append(arrs)
    return rate(*(a[:, :K] for a in arrs))
def carved(gb_tenths):
    buf = torch.zeros(int(gb_tenths * (1 << 30) / 10) // 4, dtype=torch.float32, device=dev)
    keep.append(buf)
    out = []
    n_pos = max(1, (buf.numel() * 4 - 3 * A) // (1 << 30) + 1)
    for p in range(min(n_pos, 4)):
        def view(off):
            o = ((p << 30) + off) // 4
            return buf[o:o + words].view(N, pitch)[:, :K]
        out.append(rate(view(0), view(A), view(2 * A)))
    return out
for rnd in range(3):
    print(f"round {rnd}: separate " + " ".join(f"{separate():.2f}" for _ in range(4)), flush=True)
    for g in (12, 20, 40, 80):
        print(f"          one allocation of {g / 10:.1f} GB: " + " ".join(f"{r:.2f}" for r in carved(g)), flush=True)
print("final: separate " + " ".join(f"{separate():.2f}" for _ in range(6)))
