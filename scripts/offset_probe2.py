"""Follow-up of offset_probe.py: inside ONE large allocation, how does the rate of the probe depend on the DISTANCE between the arrays?
(1) W alone at several positions; (2) W + m at distance D; (3) W + m + v at distances (D, 2 D).   python scripts/offset_probe2.py [N K GB]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K, GB = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 20000, 12)
pitch = ops.pad32(K)
words = N * pitch
nbytes = words * 4
A = (nbytes + (1 << 21) - 1) >> 21 << 21
buf = torch.zeros((GB << 30) // 4, dtype=torch.float32, device=dev)
def view(off_bytes):
    o = off_bytes // 4
    return buf[o:o + words].view(N, pitch)[:, :K]
def rate(*arrs):
    n = sum(a is not None for a in arrs)
    return 8.0 * n * N * K / ops.placement_probe_us(*arrs) / 1e6
MB = 1 << 20
print("W alone at position (MB): " + " ".join(f"{p}:{rate(view(p * MB), None, None):.2f}" for p in (0, 64, 512, 1024, 2048, 4096, 6000)), flush=True)
step = 32 * MB
Ds = [A + k * step for k in range(0, ((GB << 30) - 3 * A) // (2 * step))]
print("W + m at distance D (MB: TB/s)")
line = []
for D in Ds:
    line.append(f"{D // MB}:{rate(view(0), view(D), None):.2f}")
print(" ".join(line), flush=True)
print("W + m + v at distances (D, 2 D)")
line = []
for D in Ds:
    line.append(f"{D // MB}:{rate(view(0), view(D), view(2 * D)):.2f}")
print(" ".join(line), flush=True)
