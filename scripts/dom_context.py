"""Why does the dominant kernel take ~500 us inside the step and ~450 us in a back-to-back loop?  Same launch, different contexts:
(a) 20 launches back to back, total / 20;  (b) back to back, one event pair per launch;  (c) a ~150 us spin kernel on the same
stream between launches (the step's narrow chain: the chip is busy but HBM idle);  (d) 2 ms of host idle between launches;
(e) two weights alternating (the step updates gex then cnv)."""
import sys, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops

dev = torch.device("cuda:0")
n_out, k_in, B = 5000, 20000, 128
g = torch.Generator(device=dev); g.manual_seed(1)
ldw = ops.pad32(k_in)
ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
S = ops.dw_adam_fwd_slabs(n_out, k_in)
slabs = torch.zeros(S, B, n_out, device=dev)
dy = torch.randn(B, n_out, generator=g, device=dev) * 1e-2
x = torch.randn(B, k_in, generator=g, device=dev); xn = torch.randn(B, k_in, generator=g, device=dev)
dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, xn)
Ws = []
for _ in range(2):
    W = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
    m = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-3
    v = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-5
    Ws.append((W, m, v))


def launch(i=0):
    W, m, v = Ws[i]
    ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs)


def per_launch(n, between=None, alt=False):
    evs = []
    for i in range(n):
        if between:
            between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(i & 1 if alt else 0); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return f"median {t[len(t) // 2]:6.1f}  min {t[0]:6.1f}  max {t[-1]:6.1f} us"


for _ in range(5):
    launch()
torch.cuda.synchronize()
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record(); torch.cuda.synchronize()
    print(f"(a) back to back, total / 20:                 {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
    print(f"(b) back to back, event pair per launch:      {per_launch(20)}")
    spin = lambda: torch.cuda._sleep(300_000)          # ~ 150 us at 2 GHz
    print(f"(c) ~150 us spin kernel before each launch:   {per_launch(20, spin)}")
    print(f"(d) 2 ms host idle before each launch:        {per_launch(20, lambda: (torch.cuda.synchronize(), time.sleep(0.002)))}")
    print(f"(e) two weights alternating, back to back:    {per_launch(20, None, True)}")
    narrow = lambda: [ops.fill(ops.IMMEDIATE, ctrl[32:48], 0.0) for _ in range(12)]
    print(f"(f) 12 tiny launches before each launch:      {per_launch(20, narrow)}")
