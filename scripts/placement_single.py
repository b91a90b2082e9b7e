"""Is the placement quality of a wide weight a property of each ARRAY on its own?  Allocate N arrays one by one (spacers in between),
probe each alone, then probe triples made of the fastest three, the slowest three, and mixed ones.   python scripts/placement_single.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
out, fin = 5000, 20000
N = int(sys.argv[1]) if len(sys.argv) > 1 else 14
spacer_mb = (0, 6, 3, 254, 1201, 5, 777, 2403, 30, 3607, 2, 333, 4811, 14, 100, 6005, 62, 1022)
arrs = []
for t in range(N):
    sp = torch.empty(spacer_mb[t % len(spacer_mb)] << 20, dtype=torch.uint8, device=dev) if t else None
    a = torch.zeros(out, fin, device=dev)
    us = ops.placement_probe_us(a)
    arrs.append((us, a))
    del sp
    torch.cuda.empty_cache()
arrs.sort(key=lambda x: x[0])
print("single-array probes (us):", [round(u, 1) for u, _ in arrs])
def tri(ix):
    a, b, c = (arrs[i][1] for i in ix)
    return ops.placement_probe_us(a, b, c)
print("triple of the 3 fastest:", round(tri((0, 1, 2)), 1), " next 3:", round(tri((3, 4, 5)), 1))
print("triple of the 3 slowest:", round(tri((N - 1, N - 2, N - 3)), 1))
print("2 fastest + slowest:", round(tri((0, 1, N - 1)), 1), " fastest + 2 slowest:", round(tri((0, N - 1, N - 2)), 1))
print("orderings of (fast, fast, slow):", round(tri((N - 1, 0, 1)), 1), round(tri((0, N - 1, 1)), 1))
