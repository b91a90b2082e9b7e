"""Where the narrow chain's microseconds go: phase stamps of fx_enc_tail_fwd / fx_fusion_fwd / fx_heads_step / fx_block_bwd inside ONE replayed
step of a bench configuration (the -DFX_CHAIN_PROFILE build: thread 0 of every workgroup writes the 100 MHz wall clock at phase boundaries).
    python scripts/build_variant.py cprof fx_enc_tail.hip fx_block_bwd.hip fx_heads.hip -DFX_CHAIN_PROFILE          (build host)
    FXHIP_LIB=build_tmp/libfxhip_cprof.so python scripts/chain_profile.py [cfg2]                                     (GPU box)
Per kernel: for every phase boundary k, the median and the maximum over the workgroups of (stamp k - the launch's earliest entry stamp), us."""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from flexynesis_amd import _lib

config = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
rec, pipe, store, run = bench._engine_leg(config, 128, dev, "bf16x3", 20, 5, 1e-3)
SLOTS, WGS = 16, 1024
readers = {}
for name in ("tail", "heads", "bb"):
    fn = getattr(_lib.lib, "fx_debug_chain_stamps_" + name)
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    readers[name] = fn
buf = (C.c_ulonglong * (SLOTS * WGS))()


def read(name, reset):
    assert readers[name](buf, reset) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(WGS, SLOTS).astype(np.int64).copy()


run(4)
torch.cuda.synchronize()
for n in readers:
    read(n, 1)
run(1)
torch.cuda.synchronize()
T = {n: read(n, 0) for n in readers}
t0 = min(int(a[a[:, 0] > 0, 0].min()) for a in T.values() if (a[:, 0] > 0).any())
LABELS = {
    "enc_tail_fwd": ("tail", slice(0, 512), ["entry", "slabs summed, x stored", "batch statistics", "affine / act / dropout, out stored", "layer_out partial products", "partials stored"]),
    "fusion_fwd": ("tail", slice(512, 1024), ["entry", "column-block partials summed (ecat)", "fusion Linear, emb stored"]),
    "heads_step": ("heads", slice(0, 1024), ["entry", "forward", "loss + output gradient", "saved tensors re-read (requests)", "backward prefix", "BatchNorm backward", "embedding gradient share", "meet + total"]),
    "block_bwd": ("bb", slice(0, 1024), ["entry", "x / out loaded", "upstream products (da, gW)", "da transposed", "gate + BatchNorm backward", "dy stored", "dyT split stored", "Gram norm share"]),
}
print(f"{config}: {rec['ms_per_step']} ms/step; one replayed step, stamps relative to the first stamp of the step (us)")
for kname, (tu, rows, labels) in LABELS.items():
    a = T[tu][rows]
    live = a[:, 0] > 0
    if not live.any():
        continue
    a = a[live]
    first = int(a[:, 0].min())
    print(f"{kname}: {len(a)} workgroups, first entry at +{(first - t0) / 100:.1f} us, last entry +{(int(a[:, 0].max()) - first) / 100:.1f} us after it")
    for k, lab in enumerate(labels):
        col = a[:, k]
        ok = col > 0
        if not ok.any():
            continue
        d = (col[ok] - first) / 100.0
        print(f"    {k} {lab:42s} median {np.median(d):6.1f}   max {d.max():6.1f}   (n={int(ok.sum())})")
    if kname == "heads_step":
        for w in range(min(len(a), 4)):
            print("    workgroup", w, "(chain role)" if w % 2 == 0 else "(weight-gradient role)", [round((int(v) - first) / 100.0, 1) if v > 0 else None for v in a[w, :8]])
