"""Does the RELATIVE alignment of W, m and v decide the stream rate?  The fused kernel reads the same (row, column) of the three arrays at the same
time; separately allocated arrays all start on a 2 MiB boundary, i.e. the three streams run with identical low address bits.  Here the three live in
ONE allocation, m and v shifted by d and 2 d bytes; fx_placement_probe rates every d, on several base allocations.
python scripts/offset_probe.py [N K bases]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K, bases = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 20000, 4)
pitch = ops.pad32(K)
nbytes = N * pitch * 4
A = (nbytes + (1 << 21) - 1) >> 21 << 21                      # array stride: whole 2 MiB pages
ds = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 699136, 1048576, 1398272]
keep = []
print("base " + " ".join(f"{d:>8d}" for d in ds))
for b in range(bases):
    buf = torch.zeros((3 * A + 4 * (1 << 21)) // 4, dtype=torch.float32, device=dev)
    keep.append(buf)
    row = []
    for d in ds:
        def view(off_bytes):
            o = off_bytes // 4
            return buf[o:o + N * pitch].view(N, pitch)[:, :K]
        W, m, v = view(0), view(A + d), view(2 * A + 2 * d)
        row.append(24.0 * N * K / ops.placement_probe_us(W, m, v) / 1e6)
    print(f"{b:4d} " + " ".join(f"{r:8.2f}" for r in row), flush=True)
# the same three arrays as separate allocations, for reference
sep = []
for b in range(bases):
    arrs = [torch.zeros(N, pitch, device=dev) for _ in range(3)]
    keep.append(arrs)
    W, m, v = (a[:, :K] for a in arrs)
    sep.append(24.0 * N * K / ops.placement_probe_us(W, m, v) / 1e6)
print("separate allocations: " + " ".join(f"{r:5.2f}" for r in sep))
