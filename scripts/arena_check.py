"""PartitionArena (engine.py): build it, then rate W / m / v of several wide-weight shapes taken from it against separately allocated arrays.
python scripts/arena_check.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
from flexynesis_amd.engine import PartitionArena
dev = torch.device("cuda:0")
t0 = time.time()
ar = PartitionArena.get(dev)
print("arena:", None if ar is None else ar.info, f"({time.time() - t0:.2f} s)", PartitionArena._arenas[0].info if ar is None else "")
shapes = [(5000, 20000), (7500, 30016), (8879, 19904), (10000, 40000), (4000, 20000), (20000, 5024)]
keep = []
for (H, ld) in shapes:
    F = ld - (ld % 32 and 0)
    if ar is not None:
        (w, m, v), tok = ar.take3(H * ld)
        a_rate = 24.0 * H * ld / ops.placement_probe_us(w.view(H, ld), m.view(H, ld), v.view(H, ld)) / 1e6
        keep.append((w, m, v))            # (the ranges stay leased while these tensors live)
    else:
        a_rate = float("nan")
    seps = []
    for _ in range(3):
        arrs = [torch.zeros(H, ld, device=dev) for _ in range(3)]
        keep.append(arrs)
        seps.append(24.0 * H * ld / ops.placement_probe_us(*arrs) / 1e6)
    print(f"[{H}, {ld}]: arena {a_rate:.2f} TB/s | separate allocations " + " ".join(f"{r:.2f}" for r in seps), flush=True)
