"""Two arrays that are slow alone can be fast together (scripts/prefix_rating2.py): is memory made of partitions that each deliver ~4.9 TB/s to this access
pattern?  One allocation of G GB; W fixed at offset 0, m at offset y: the pair is rated for y = 0.5 GB .. G GB.   python scripts/partition_map.py [G step_GB]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
step = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
N, K = 5000, 20000
words = N * K
buf = torch.empty((G << 30) // 4, dtype=torch.float32, device=dev)
def view(off_bytes):
    o = int(off_bytes) // 4 // 64 * 64
    return buf[o:o + words].view(N, K)
def pair(x, y):
    return 16.0 * N * K / ops.placement_probe_us(view(x), view(y), None) / 1e6
def single(x):
    return 8.0 * N * K / ops.placement_probe_us(view(x), None, None) / 1e6
GBb = 1 << 30
ys = [0.5 + step * i for i in range(int((G - 1) / step))]
print("single array at y (GB): " + " ".join(f"{y:g}:{single(y * GBb):.2f}" for y in ys), flush=True)
print("pair (0, y): " + " ".join(f"{y:g}:{pair(0, y * GBb):.2f}" for y in ys), flush=True)
