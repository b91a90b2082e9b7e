"""Fraction of FAST placements of a wide weight's W / m / v as a function of the row pitch (the fused dW + Adam traversal, rated by
fx_placement_probe: no GEMM, same addresses).  Every trial allocates fresh arrays while all earlier ones of the pitch stay allocated, so
each trial is another piece of physical memory.   python scripts/pitch_sweep.py N K trials pitch [pitch ...]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K, trials = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for pitch in [int(a) for a in sys.argv[4:]]:
    rates, keep = [], []
    for trial in range(trials):
        arrs = [torch.zeros(N, pitch, device=dev) for _ in range(3)]
        keep.append(arrs)
        W, m, v = (a[:, :K] for a in arrs)
        rates.append(24.0 * N * K / ops.placement_probe_us(W, m, v) / 1e6)
    del keep, arrs, W, m, v
    torch.cuda.empty_cache()
    fast = sum(r >= 5.6 for r in rates)
    print(f"pitch {pitch:6d} floats  {pitch * 4 % 4096:4d} mod 4096 B   fast {fast:2d}/{trials}   TB/s " + " ".join(f"{r:4.2f}" for r in rates), flush=True)
