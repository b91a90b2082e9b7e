"""Follow-up of prefix_rating.py: row RANGES of separately allocated [10000, 20000] arrays rated alone -- is an array fast because part of it lies in
fast memory, or because its parts lie in DIFFERENT memory?   python scripts/prefix_rating2.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
F, Hmax = 20000, 10000
rngs = [(0, 10000), (0, 5000), (5000, 10000), (2500, 7500), (0, 2500), (2500, 5000), (5000, 7500), (7500, 10000)]
keep = []
print("array  " + " ".join(f"{a:5d}:{b:<5d}" for a, b in rngs))
for a in range(14):
    arr = torch.zeros(Hmax, F, device=dev)
    keep.append(arr)
    row = [8.0 * (h1 - h0) * F / ops.placement_probe_us(arr[h0:h1], None, None) / 1e6 for h0, h1 in rngs]
    print(f"{a:5d}  " + " ".join(f"{r:11.2f}" for r in row), flush=True)
# two arrays that are slow alone, streamed together (W + m): does the pair run faster than either?
slow = [k for k in keep if 8.0 * Hmax * F / ops.placement_probe_us(k, None, None) / 1e6 < 5.1]
for i in range(0, len(slow) - 1, 2):
    print("pair of slow arrays:", round(16.0 * Hmax * F / ops.placement_probe_us(slow[i], slow[i + 1], None) / 1e6, 2))
