"""The M = 128 forward products (cfg3's decoder FC_output [128 x 5000] -> 20000; the unfused encoder forward [128 x 20000] -> 5000) under the
kernel choices of fx_linear_fwd_bf16x3_ex: the 128 x 128-tile kernel (shipped for 65..128 rows) against the register-fragment kernel
(no_mt = 4: four waves of 32 rows, X fragments straight into registers), in the parity mode (bf16x3) and the plain-bf16 mode.
    python scripts/fwd128_variants.py"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
ws = ops.Workspace(dev)
for (M, K, N) in ((128, 5000, 20000), (128, 20000, 5000), (100, 20000, 5000)):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.01
    b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev)
    sp = ops.new_split_kb(M, K, dev)
    ops.split_bf16(ops.IMMEDIATE, sp[0], sp[1], x)
    ref = None
    for products in (3, 1):
        for no_mt, wc in ((0, 0), (4, 0), (0, 8)):
            for splitk in (0, 2, 3, 4, 6):
                rec = ops.TapeRecorder(products=products)
                need = max(splitk, 1) * M * N * 4 if splitk else int(ops.lib.fx_linear_fwd_bf16x3_workspace_bytes(M, N, K))
                ws.reserve(max(need, 8 * M * N * 4))
                rec.emit("fx_linear_fwd_bf16x3_ex", y.data_ptr(), sp[0].data_ptr(), ops._lo(rec, sp[1]), W.data_ptr(), b.data_ptr(), M, N, K,
                         sp[0].shape[1], W.stride(0), y.stride(0), ws.buf.data_ptr(), ws.nbytes, splitk, wc, no_mt, 0)
                try:
                    rec.run(); torch.cuda.synchronize()
                except Exception as e:
                    print(f"[{M} x {K}] -> {N} products {products} no_mt {no_mt} wave_cols {wc} splitk {splitk}: {e}")
                    continue
                if products == 3 and ref is None:
                    ref = y.clone()
                err = float((y - ref).abs().max() / ref.abs().max())
                for _ in range(5):
                    rec.run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    rec.run()
                e1.record(); e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 50
                print(f"[{M} x {K}] -> {N}  products {products}  kernel {'reg-fragment' if no_mt == 4 else ('128x256 tile ' if wc == 8 else '128x128 tile ')}  splitk {splitk or 'auto'}: "
                      f"{us:7.1f} us (incl. slab reduce)  W at {4.0 * N * K / us / 1e6:5.2f} TB/s   max dev vs bf16x3 {err:.1e}", flush=True)
