"""Partition structure inside one allocation of G GB: pair ratings for a grid of offsets, a finer scan around the first boundary, triples, and what
the allocation itself costs.   python scripts/partition_map2.py [G]"""
import sys, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, K = 5000, 20000
words = N * K
torch.cuda.synchronize(); t0 = time.time()
buf = torch.empty((G << 30) // 4, dtype=torch.float32, device=dev)
torch.cuda.synchronize(); print(f"allocating {G} GB took {time.time() - t0:.3f} s", flush=True)
GBb = 1 << 30
def view(off_gb):
    o = int(off_gb * GBb) // 4 // 64 * 64
    return buf[o:o + words].view(N, K)
def rate(*offs):
    arrs = [view(o) for o in offs] + [None] * (3 - len(offs))
    return 8.0 * len(offs) * N * K / ops.placement_probe_us(*arrs) / 1e6
xs = [0, 30, 60, 70, 100, 126, 130, 160, 190]
xs = [x for x in xs if x + 1 < G]
print("pairs:      " + " ".join(f"{x:>6g}" for x in xs))
for a in xs:
    print(f"{a:>6g}      " + " ".join(f"{rate(a, b):6.2f}" if a != b else "     -" for b in xs), flush=True)
print("boundary scan, pair (0, y): " + " ".join(f"{y:g}:{rate(0, y):.2f}" for y in [62 + 0.25 * i for i in range(13)]), flush=True)
for t in ((0, 70, 130), (0, 70, 100), (0, 30, 70), (0, 70, 190), (0, 130, 190), (0, 1, 2)):
    if max(t) + 1 < G:
        print("triple", t, f"{rate(*t):.2f}", flush=True)
