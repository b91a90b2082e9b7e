"""Host -> HBM strategies for a [2048, 20000] fp32 matrix: pageable .to() (what DeviceImporter.upload does), a pinned
double-buffer staging loop, in-place hipHostRegister + one DMA, and an already pinned source."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd.ingest import DeviceImporter

N, F = 2048, 20000
x = np.random.default_rng(0).normal(size=(N, F)).astype(np.float32)
mb = x.nbytes / 1e6
imp = DeviceImporter()

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

a = t(lambda: torch.from_numpy(x).to("cuda"))
print(f"pageable .to():        {a*1e3:7.2f} ms  {mb/a/1e3:6.2f} GB/s")
b = t(lambda: imp.upload(x))
print(f"DeviceImporter.upload: {b*1e3:7.2f} ms  {mb/b/1e3:6.2f} GB/s")
pins = [torch.empty(64 << 20, dtype=torch.uint8).pin_memory() for _ in range(2)]
def staged():
    src = torch.from_numpy(x); dst = torch.empty((N, F), dtype=torch.float32, device="cuda")
    rows_per = (64 << 20) // (F * 4); ev = [None, None]
    for k, r0 in enumerate(range(0, N, rows_per)):
        r1 = min(N, r0 + rows_per); b_ = k & 1
        if ev[b_] is not None: ev[b_].synchronize()
        st = pins[b_][: (r1 - r0) * F * 4].view(torch.float32).view(r1 - r0, F)
        st.copy_(src[r0:r1]); dst[r0:r1].copy_(st, non_blocking=True)
        ev[b_] = torch.cuda.Event(); ev[b_].record()
    return dst
b2 = t(staged)
print(f"pinned double buffer:  {b2*1e3:7.2f} ms  {mb/b2/1e3:6.2f} GB/s")
rt = torch.cuda.cudart()
def reg():
    src = torch.from_numpy(x)
    rc = rt.cudaHostRegister(src.data_ptr(), src.numel() * 4, 0)
    assert int(rc) == 0, rc
    try:
        d = torch.empty((N, F), dtype=torch.float32, device="cuda")
        d.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
    finally:
        rt.cudaHostUnregister(src.data_ptr())
    return d
c = t(reg)
print(f"hostRegister + DMA:    {c*1e3:7.2f} ms  {mb/c/1e3:6.2f} GB/s   is_pinned-after-register:", end=" ")
src = torch.from_numpy(x); rt.cudaHostRegister(src.data_ptr(), src.numel()*4, 0); print(src.is_pinned()); rt.cudaHostUnregister(src.data_ptr())
xp = torch.from_numpy(x).pin_memory()
d = t(lambda: xp.to("cuda", non_blocking=True))
print(f"already pinned DMA:    {d*1e3:7.2f} ms  {mb/d/1e3:6.2f} GB/s")
