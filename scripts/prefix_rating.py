"""Does a fast array stay fast for the row-prefix views an HPO sweep would take from it (same pitch, H = int(F x U[0.2, 0.5]) rows)?
16 separately allocated [10000, 20000] arrays, each rated alone (fx_placement_probe, W only) as [H, 20000] for several H.   python scripts/prefix_rating.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
F, Hmax = 20000, 10000
Hs = [10000, 9000, 8000, 7000, 6000, 5000, 4000]
keep = []
print("array  " + " ".join(f"H={h:5d}" for h in Hs))
for a in range(16):
    arr = torch.zeros(Hmax, F, device=dev)
    keep.append(arr)
    row = [8.0 * h * F / ops.placement_probe_us(arr[:h], None, None) / 1e6 for h in Hs]
    print(f"{a:5d}  " + " ".join(f"{r:7.2f}" for r in row), flush=True)
