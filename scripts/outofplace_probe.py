"""does an OUT-OF-PLACE (ping-pong) update stream faster than the in-place one?  The dominant kernel reads AND
writes W / m / v; in place every memory partition sees both directions.  fx_placement_probe_oop reads three arrays and writes three
others with the fused kernel's exact tile schedule; the arrays come from the partition arena's pools (A = the partition the process
stands in, B1 / B2 = the partitions behind the boundaries it found), so every (source partitions -> destination partitions) layout
can be rated.      python scripts/outofplace_probe.py [H F]     (default 5000 20000; prints a table, TB/s = 24 B per element)"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
from flexynesis_amd.engine import PartitionArena

H, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 20000)
dev = torch.device("cuda:0")
ar = PartitionArena.get(dev)
assert ar is not None, PartitionArena._arenas[0].info
print("arena:", {k: ar.info.get(k) for k in ("pool_GB", "pool_B_classes", "spacer_GB", "build_s", "pool_A_ends_TBps")})
ld = (F + 31) // 32 * 32
nbytes = H * ld * 4
kinds = sorted(set(ar.kind))
names = {0: "A", 1: "B1", 2: "B2"}


def take(kind):
    with ar.lock:
        r = ar._alloc(kind, nbytes)
    assert r is not None, f"pool {names[kind]} exhausted"
    ci, off, _ = r
    t = ar.chunks[ci][off:off + nbytes].view(torch.float32).view(H, ld)[:, :F]
    t.zero_()
    return t


arr = {k: [take(k) for _ in range(6 if k == 0 else 6)] for k in kinds}
torch.cuda.synchronize()
rate = lambda us: 24.0 * H * F / us / 1e6
rows = []


def inplace(tag, w, m, v):
    us = ops.placement_probe_us(w, m, v, launches=5)
    rows.append((f"in place   {tag}", us, rate(us)))


def oop(tag, src, dst):
    us = ops.placement_probe_oop_us(dst, src, launches=5)
    rows.append((f"out of place {tag}", us, rate(us)))


A, B1 = arr[0], arr[1]
B2 = arr.get(2)
inplace("W A | m A | v A        (one partition: the slow placement)", A[0], A[1], A[2])
inplace("W A | m B1 | v B1      (the arena's layout)", A[0], B1[0], B1[1])
if B2:
    inplace("W A | m B1 | v B2      (three partitions)", A[0], B1[0], B2[0])
oop("A,A,A -> B1,B1,B1          (pure-read / pure-write partitions)", (A[0], A[1], A[2]), (B1[0], B1[1], B1[2]))
oop("A,B1,B1 -> B1,A,A          (swap: both partitions read and write)", (A[0], B1[0], B1[1]), (B1[2], A[1], A[2]))
oop("A,A,B1 -> B1,B1,A", (A[0], A[1], B1[0]), (B1[1], B1[2], A[2]))
oop("A,A,A -> A,A,A             (other arrays of the same partition)", (A[0], A[1], A[2]), (A[3], A[4], A[5]))
if B2:
    oop("A,A,A -> B1,B1,B2", (A[0], A[1], A[2]), (B1[0], B1[1], B2[0]))
    oop("A,B1,B1 -> B2,B2,B2        (reads from two, writes to a third)", (A[0], B1[0], B1[1]), (B2[0], B2[1], B2[2]))
    oop("A,B1,B2 -> B1,B2,A         (rotate over three)", (A[0], B1[0], B2[0]), (B1[1], B2[1], A[1]))
# the plain copy kernel for scale: 1.2 GB read + 1.2 GB written
src = torch.empty(3 * H * ld, device=dev)
dst = torch.empty_like(src)
ops.stream_copy(ops.IMMEDIATE, dst, src)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.stream_copy(ops.IMMEDIATE, dst, src)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 5
rows.append(("fx_stream_copy of the same bytes (torch allocations)", us, 8.0 * src.numel() / us / 1e6))
print(f"[{H}, {F}] fp32, three arrays, fused-kernel tile schedule (512 workgroups, 2 per CU)")
for tag, us, tbs in rows:
    print(f"  {us:8.1f} us  {tbs:5.2f} TB/s   {tag}")
