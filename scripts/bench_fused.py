"""Stand-alone timing of the dominant kernel fx_linear_dw_adam_fwd_bf16x3 at the cfg2 shape (HIP events), one line per
``flags`` variant (default: 1 = non-temporal W / m / v; more via FX_BENCH_FLAGS=0x3,0x601,...), plus a bit-identity check
of W / m / v / slabs between the variants.  Usage: python scripts/bench_fused.py [n_launches] [H] [F]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd import ops
from flexynesis_amd._lib import lib
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
F = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
B = 128
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn(B, F, device=dev, generator=g)
Xn = torch.randn(B, F, device=dev, generator=g)
dY = torch.randn(B, H, device=dev, generator=g) * 1e-3
ldw = ops.pad32(F)
def fresh():
    gg = torch.Generator(device=dev); gg.manual_seed(1)
    W = torch.randn(H, ldw, device=dev, generator=gg) * 0.01
    return W, torch.zeros_like(W), torch.zeros_like(W)
xt = ops.new_split(F, B, dev); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], X)
dyt = ops.new_split(H, B, dev); ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dY)
xn = ops.new_split_kb(B, F, dev); ops.split_bf16(ops.IMMEDIATE, xn[0], xn[1], Xn)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
S = 24          # slab buffers sized for the largest run count any variant asks for (flags bits 8-15)
st = torch.cuda.current_stream().cuda_stream
def launch(W, M, V, slabs, flags):
    rc = lib.fx_linear_dw_adam_fwd_bf16x3(W.data_ptr(), M.data_ptr(), V.data_ptr(), dyt[0].data_ptr(), dyt[1].data_ptr(),
                                          xt[0].data_ptr(), xt[1].data_ptr(), B, H, F, dyt[0].stride(0), xt[0].stride(0), ldw,
                                          ctrl.data_ptr(), xn[0].data_ptr(), xn[1].data_ptr(), xn[0].shape[1], B,
                                          slabs.data_ptr(), slabs.numel() * 4, flags, st)
    assert rc == 0, rc
variants = [("default (nt)", 1)]
extra = os.environ.get("FX_BENCH_FLAGS", "")
for tok in extra.split(","):
    if tok:
        variants.append((f"flags {tok}", int(tok, 0)))
outs = {}
for name, fl in variants:
    W, M, V = fresh()
    slabs = torch.zeros(S, B, H, device=dev)
    for _ in range(3):
        launch(W, M, V, slabs, fl)
    torch.cuda.synchronize()
    outs[name] = (W.clone(), M.clone(), V.clone(), slabs.clone())
base = outs[variants[0][0]]
for name, _ in variants:
    same = all(torch.equal(a, b) for a, b in zip(outs[name][:3], base[:3]))
    ysum, ybase = outs[name][3].sum(0), base[3].sum(0)          # (the slab COUNT differs between variants: compare the reduced forward)
    yerr = float((ysum - ybase).abs().max() / ybase.abs().max())
    print(f"[bit-identity] {name} vs default: W/m/v {'identical' if same else 'DIFFERENT'}, forward max rel diff {yerr:.2e}", flush=True)
del outs
# two weight sets so consecutive launches do not hit a warm Infinity Cache
sets = [fresh() + (torch.zeros(S, B, H, device=dev),) for _ in range(2)]
for rep in range(2):
    for name, fl in variants:
        for i in range(2):
            launch(*sets[i], fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            launch(*sets[i % 2], fl)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"[{H}x{F} B={B} S={S}] {name:24s} {us:7.1f} us/launch = {24.0 * H * F / us / 1e6:.2f} TB/s = {24.0 * H * F / us / 1e6 / 8:.3f} of 8 TB/s", flush=True)
