"""Inside ONE large allocation (where every position of a pitch-20000 array rates slow, scripts/offset_probe2.py): does the row PITCH change the rating?
W / m / v back to back at each pitch; the same pitch is also rated at a second position of the allocation.   python scripts/offset_probe3.py [N K GB]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K, GB = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 20000, 8)
buf = torch.zeros((GB << 30) // 4, dtype=torch.float32, device=dev)
base = ops.pad32(K)
def rate_at(pitch, start_bytes):
    words = N * pitch
    A = (words * 4 + (1 << 21) - 1) >> 21 << 21
    def view(off):
        o = (start_bytes + off) // 4
        return buf[o:o + words].view(N, pitch)[:, :K]
    return 24.0 * N * K / ops.placement_probe_us(view(0), view(A), view(2 * A)) / 1e6
out = []
for k in range(0, 48):
    pitch = base + 32 * k
    out.append((pitch, rate_at(pitch, 0), rate_at(pitch, 3 << 30)))
print(" ".join(f"{p}:{a:.2f}/{b:.2f}" for p, a, b in out))
print("fast (>= 5.6) at position 0:", sum(a >= 5.6 for _, a, _ in out), "of", len(out), "| at 3 GiB:", sum(b >= 5.6 for _, _, b in out))
