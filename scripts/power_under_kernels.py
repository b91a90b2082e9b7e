"""Package power and shader clock (rocm-smi, 0.5 s samples) beside 5 s loops of the wide kernels: the GEMM-free placement probe, the fused dominant kernel
(cfg2 shape), the K = 384 dW + Adam launch (cfg4 shape), the unfused forward at M = 128 and M = 384 -- each in the parity mode (split bf16, three
products) and, round 6, in the plain-bf16 throughput mode (one product: the `lo` operands are NULL).   python scripts/power_under_kernels.py"""
import subprocess, sys, threading, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
from flexynesis_amd.engine import PartitionArena
dev = torch.device("cuda:0")
ar = PartitionArena.get(dev)
def arrays(N, K):
    ld = ops.pad32(K)
    if ar is not None:
        (w, m, v), _ = ar.take3(N * ld)
        return [t.view(N, ld)[:, :K] for t in (w, m, v)]
    return [torch.zeros(N, ld, device=dev)[:, :K] for _ in range(3)]
def sample(fn, label, secs=5.0):
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw = [l for l in out.splitlines() if "Power" in l and "W" in l]
                ck = [l for l in out.splitlines() if "sclk" in l]
                samples.append((pw[0].split(":")[-1].strip() if pw else "?", ck[0].split(":")[-1].strip() if ck else "?"))
            except Exception as e:
                samples.append((repr(e)[:40], ""))
            time.sleep(0.5)
    th = threading.Thread(target=sampler); th.start()
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); n = 0; e0.record()
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop[0] = True; th.join()
    print(f"{label:58s} {e0.elapsed_time(e1) / n * 1e3:7.1f} us per launch   " + " | ".join(f"{p} {c}" for p, c in samples[2:8]), flush=True)
def plain(op, *a, **k):
    """The same wrapper call recorded on a plain-bf16 tape (products = 1) -> a callable that re-issues it."""
    rec = ops.TapeRecorder(products=1)
    op(rec, *a, **k)
    return rec.run
ctrl = torch.zeros(64, device=dev); ctrl[0] = 9.0
ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3); ctrl[4] = 0.5
# cfg2 shape
N, K, B = 5000, 20000, 128
W, m, v = arrays(N, K); W.normal_(0, 0.01)
dy = torch.randn(B, N, device=dev) * 1e-2; x = torch.randn(B, K, device=dev)
dyt, xt = ops.new_split(N, B, dev), ops.new_split(K, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, K, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, x)
slabs = torch.zeros(16, B, N, device=dev)
sample(lambda: ops.IMMEDIATE.emit("fx_placement_probe", W.data_ptr(), m.data_ptr(), v.data_ptr(), N, K, W.stride(0)), "placement probe [5000, 20000] (no GEMM)")
sample(lambda: ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs), "fused dW + Adam + forward [5000, 20000], B = 128")
sample(lambda: ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl), "unfused dW + Adam [5000, 20000], K = 128")
Y = torch.empty(B, N, device=dev); ws = ops.Workspace(dev); b = torch.zeros(N, device=dev)
sample(lambda: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, xnh, xnl, W, b, ws), "unfused forward [128 x 20000] -> 5000")
sample(plain(ops.linear_dw_adam_fwd_bf16x3, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs), "  plain bf16: fused dW + Adam + forward [5000, 20000], B = 128")
sample(plain(ops.linear_dw_adam_bf16x3, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl), "  plain bf16: unfused dW + Adam [5000, 20000], K = 128")
sample(plain(ops.linear_fwd_bf16x3, Y, xnh, xnl, W, b, ws), "  plain bf16: unfused forward [128 x 20000] -> 5000")
# the VAE decoders' FC_output [20000, 5000]: forward [128 x 5000] -> 20000 and its data gradient (cfg3's four HBM-bound passes)
Wd = torch.randn(20000, 5000, device=dev) * 0.01
hsp = ops.new_split_kb(B, 5000, dev); ops.split_bf16(ops.IMMEDIATE, hsp[0], hsp[1], torch.randn(B, 5000, device=dev))
Yd = torch.empty(B, 20000, device=dev); bd = torch.zeros(20000, device=dev)
dsp = ops.new_split_kb(B, 20000, dev); ops.split_bf16(ops.IMMEDIATE, dsp[0], dsp[1], torch.randn(B, 20000, device=dev) * 1e-3)
dxd = torch.empty(B, 5000, device=dev)
sample(lambda: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Yd, hsp[0], hsp[1], Wd, bd, ws), "decoder forward [128 x 5000] -> 20000 (cfg3 FC_output)")
sample(lambda: ops.linear_bwd_x_bf16x3(ops.IMMEDIATE, dxd, dsp[0], dsp[1], Wd, ws), "decoder data gradient [128 x 20000] . W[20000, 5000]")
sample(plain(ops.linear_fwd_bf16x3, Yd, hsp[0], hsp[1], Wd, bd, ws), "  plain bf16: decoder forward")
sample(plain(ops.linear_bwd_x_bf16x3, dxd, dsp[0], dsp[1], Wd, ws), "  plain bf16: decoder data gradient")
del Wd, Yd, dxd
# cfg4 shape
N, K, B = 7500, 30000, 384
W, m, v = arrays(N, K); W.normal_(0, 0.01)
dy = torch.randn(B, N, device=dev) * 1e-2; x = torch.randn(B, K, device=dev)
dyt, xt = ops.new_split(N, B, dev), ops.new_split(K, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xs = ops.new_split_kb(B, K, dev); ops.split_bf16(ops.IMMEDIATE, xs[0], xs[1], x)
sample(lambda: ops.IMMEDIATE.emit("fx_placement_probe", W.data_ptr(), m.data_ptr(), v.data_ptr(), N, K, W.stride(0)), "placement probe [7500, 30000] (no GEMM)")
sample(lambda: ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl), "unfused dW + Adam [7500, 30000], K = 384")
Y = torch.empty(B, N, device=dev); b = torch.zeros(N, device=dev)
sample(lambda: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, xs[0], xs[1], W, b, ws), "stacked-rows forward [384 x 30000] -> 7500")
sample(plain(ops.linear_dw_adam_bf16x3, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl), "  plain bf16: unfused dW + Adam [7500, 30000], K = 384")
sample(plain(ops.linear_fwd_bf16x3, Y, xs[0], xs[1], W, b, ws), "  plain bf16: stacked-rows forward [384 x 30000] -> 7500")
